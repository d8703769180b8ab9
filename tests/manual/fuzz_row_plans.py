"""Randomised check of the row-plan GEMM dispatch (aigv_op_gemm_rows) on an MI355X - by hand, not part of the suites:

    python tests/manual/fuzz_row_plans.py [cases] [seed]

Per case: random sequence lengths (whole tiles, ragged one- and two-half tails, tiny tails, sequences shorter than a half tile, uniform and
mixed batches), N in {256, 384, 512, 1024, 1280, 4096}, K in {128, 192, 512, 1024, 2048, 3584}, every epilogue, a random setting of the
fused-tail / LONE-body knobs.  Checked: (1) against the rounded fp32 reference of the op (the suite's ulp bars); (2) every sequence alone gives the
bits it has inside the batch; (3) the knob settings give the bits of the default setting; (4) rows outside the plan's tails equal the one-kernel
full-K form (mode 2 where N is a multiple of 256) bit for bit."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_ops as T  # noqa: E402  (the suite's helpers: reference, ulp bars, the ctypes call)
from aigv_assessor_amd import native  # noqa: E402

lib = native.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def random_lens():
    kind = rng.choice(["uniform_vit", "uniform_llm", "mixed", "short", "big"])
    if kind == "uniform_vit":
        return [1025] * rng.randint(1, 9)
    if kind == "uniform_llm":
        return [rng.choice([2176, 2177, 641, 300])] * rng.randint(1, 4)
    if kind == "short":
        return [rng.randint(1, 140) for _ in range(rng.randint(1, 6))]
    if kind == "big":
        return [rng.choice([256, 512, 768, 1024]) + rng.choice([0, 0, 1, 3, 4, 5, 127, 128, 129, 255]) for _ in range(rng.randint(1, 5))]
    return [rng.choice([rng.randint(1, 700), 1025, 2176, 256, 130, 4, 5]) for _ in range(rng.randint(2, 7))]


bad = 0
for ci in range(cases):
    lens = random_lens()
    N = rng.choice([256, 384, 512, 1024, 1280, 4096])
    K = rng.choice([128, 192, 512, 1024, 2048, 3584])
    epi = rng.choice([0, 1, 2, 3, 4])
    if epi == 4 and N % 256:
        epi = 0
    if sum(lens) * max(N, K) > 6000 * 4096:
        lens = lens[:2]
    g = torch.Generator().manual_seed(ci * 131 + N + K + epi)
    A, W, bias, ls, resid, nout = T._gemm_rows_case(g, lens, N, K, epi)
    tag = f"case {ci}: lens {lens} N {N} K {K} epi {epi}"
    try:
        want = T.gemm_ref(A, W, epi, bias, ls, resid)
        base = T._run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
        T.ulp_check(base, want, frac=0.03 if epi in (1, 4) else 0.02, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)
        r0 = 0
        for n in lens:                                           # (2) alone == in the batch
            sl = slice(r0, r0 + n)
            one = T._run_gemm_rows(lib, A[sl], W, bias, ls, None if resid is None else resid[sl], nout, [n], epi)
            assert torch.equal(one.view(torch.int16), base[sl].view(torch.int16)), f"sequence of {n} rows at {r0} differs alone"
            r0 += n
        fuse, lone = rng.choice([1, 2]), rng.choice([0, 1, 2])   # (3) the launch-shape knobs move no bit
        try:
            native.check(lib.aigv_tune_default(12, fuse))
            native.check(lib.aigv_tune_default(13, lone))
            alt = T._run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
        finally:
            native.check(lib.aigv_tune_default(12, 0))
            native.check(lib.aigv_tune_default(13, 1))
        assert torch.equal(alt.view(torch.int16), base.view(torch.int16)), f"fuse_tails {fuse} / lone_body {lone} moved bits"
        if N % 256 == 0:                                         # (4) body rows == the one-kernel full-K form
            try:
                native.check(lib.aigv_tune_gemm(2, 0.0))
                full = T._run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
            finally:
                native.check(lib.aigv_tune_gemm(0, 0.0))
            r0 = 0
            for n in lens:
                nb = n // 256 * 256
                assert torch.equal(full[r0:r0 + nb].view(torch.int16), base[r0:r0 + nb].view(torch.int16)), f"body rows of the sequence at {r0} differ from the full-K kernel"
                r0 += n
        print("ok  ", tag, flush=True)
    except native.NativeError as e:                              # a shape the dispatch refuses (tiny tails need K % 128 == 0): refused loudly is fine
        if "no skinny form" not in str(e):
            bad += 1
        print("skip" if "no skinny form" in str(e) else "FAIL", tag, "->", str(e)[:200], flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", tag, "->", str(e)[:300], flush=True)
print(f"{cases - bad} / {cases} cases clean")
sys.exit(1 if bad else 0)
