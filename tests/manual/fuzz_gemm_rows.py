"""Randomised run of the scoring pass's GEMM dispatch (aigv_op_gemm_rows: per-sequence body tiles, split-K tail halves, tiny tails) at the widths the models use: random
sequence-length lists - drawn around the tile edges (1, 2, 127..129, 255..257, 383..385, 1024, 1025, 2176, 2177) and uniformly up to 3000 - x (N, K) of InternViT-300M / 6B and
InternLM2-8B / 20B x the five epilogues.  Each case: finite, within the op bar of the rounded fp32 reference, and every sequence ALONE equal to itself inside the batch bit for bit.

    python tests/manual/fuzz_gemm_rows.py [n_cases = 60] [seed = 0]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ops as T  # noqa: E402  (the suite's own helpers: case builder, launcher, reference, ulp bar)
from aigv_assessor_amd import native  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = native.load()
rng = random.Random(seed0)
EDGES = [1, 2, 3, 64, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 1024, 1025, 1026, 2176, 2177, 2175]
SHAPES = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096),                      # InternViT-300M: qkv, proj, fc1, fc2
          (6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336),                    # InternLM2-8B: wqkv, wo, w1|w3, w2
          (9600, 3200), (3200, 3200), (12800, 3200), (3200, 12800),                    # InternViT-6B
          (8192, 6144), (6144, 6144), (32768, 6144), (6144, 16384), (4096, 4096)]      # InternLM2-20B, mlp1
bad = 0
for c in range(n_cases):
    N, K = rng.choice(SHAPES)
    epi = rng.randint(0, 4)
    if epi == 4 and N % 512:
        epi = 0
    lens = [rng.choice(EDGES) if rng.random() < 0.6 else rng.randint(1, 3000) for _ in range(rng.randint(1, 5))]
    while sum(lens) * max(N, K) > 3.0e8:                                               # keep a case under ~0.6 GB per operand
        lens.pop()
    if not lens:
        lens = [rng.choice(EDGES)]
    g = torch.Generator().manual_seed(seed0 * 100000 + c)
    A, W, bias, ls, resid, nout = T._gemm_rows_case(g, lens, N, K, epi)
    both = T._run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
    ok = bool(torch.isfinite(both.float()).all())
    try:
        # (GELU: a pre-activation that lands one bf16 ulp away - fp32 summation order - moves the OUTPUT by up to ~12 ulps in the negative tail, where
        #  d ln|gelu| / dx ~ 3: over 10^7 elements per case that happens, so the worst-element bar is wider here than the suite's 4 on its fixed cases)
        T.ulp_check(both, T.gemm_ref(A, W, epi, bias, ls, resid), frac=0.03 if epi in (1, 4) else 0.02, max_ulps=16 if epi == 1 else 4 if epi == 4 else 2,
                    atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)
    except AssertionError as e:
        ok = False
        print("  reference bar:", str(e)[:200])
    r0 = 0
    for n in lens:
        sl = slice(r0, r0 + n)
        one = T._run_gemm_rows(lib, A[sl], W, bias, ls, None if resid is None else resid[sl], nout, [n], epi)
        if not torch.equal(one.view(torch.int16), both[sl].view(torch.int16)):
            ok = False
            print(f"  sequence of {n} rows at {r0}: alone != in batch")
        r0 += n
    torch.cuda.synchronize()
    T._KEEP.clear()
    if not ok:
        bad += 1
        print(f"CASE {c} FAILED: N {N} K {K} epi {epi} lens {lens}", flush=True)
    if c % 10 == 9:
        print(f"case {c + 1}/{n_cases}: failed so far {bad}", flush=True)
assert bad == 0, bad
print(f"FUZZ_GEMM_OK {n_cases} cases")
