"""What does `nt` on the GEMM epilogue's global_store_dwordx4 do on gfx950?  In-step, arm `nt` ran the four-clip step 15 % faster than the
plain store - with DIFFERENT result bits (profiles/r6_store_policy.txt), where `sc1` and `sc1 nt` were bit-identical and not faster.  This probe
separates "the stores are wrong" from "a later reader sees stale lines", op by op, on the diagnostic build (scripts/store_policy_ab.py build):

    python scripts/store_policy_probe.py          # MI355X

For each policy (0 plain, 5 nt, 6 sc1, 7 sc1 nt), one 8192 x 4096 x 4096 GEMM (EPI_STORE; then EPI_RESID in place, C == resid) into
  (a) a fresh buffer, read back by a D2H copy and by a GPU kernel;
  (b) a buffer every line of which a GPU kernel READ just before (clean copies in the L2s), then read back both ways;
and the time of the launch."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AIGV_AMD_LIB", os.path.join(ROOT, "scripts", "_abl", "libaigv_store_ab.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from aigv_assessor_amd import native  # noqa: E402
from aigv_assessor_amd.native import ptr  # noqa: E402

lib = native.load()
BF = torch.bfloat16
M, N, K = 8192, 4096, 4096
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(BF)
W = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(BF)
R = torch.randn(M, N, generator=g, device="cuda").to(BF)


def gemm(C, epi, resid=None):
    native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), N, None, None, ptr(resid), N, None, 0, M, N, K, epi, None))


def policy(v):
    native.check(lib.aigv_tune_gemm(2 + 16 * v if v else 2 + 16 * 2, 0.0))     # bits 4..6 = 1 + schedule variant; 2 = variant 1 = the shipped schedule


policy(0)
want = {}
for epi in (0, 3):
    C = R.clone() if epi == 3 else torch.empty(M, N, dtype=BF, device="cuda")
    gemm(C, epi, C if epi == 3 else None)
    torch.cuda.synchronize()
    want[epi] = C.clone()

for v in (0, 5, 6, 7):
    policy(v)
    for epi in (0, 3):
        res = []
        for pre_read in (False, True):
            C = R.clone() if epi == 3 else torch.full((M, N), 7.0, dtype=BF, device="cuda")
            torch.cuda.synchronize()
            if pre_read:
                _ = (C.float().sum() + C.float().abs().max()).item()        # two kernels read every line of C through the L2s
            gemm(C, epi, C if epi == 3 else None)
            torch.cuda.synchronize()
            bad_gpu = int((C != want[epi]).sum().item())                    # a GPU kernel reads C (through the L2s)
            bad_d2h = int((C.cpu() != want[epi].cpu()).sum().item())        # the copy engine / blit reads C
            res.append((bad_gpu, bad_d2h))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        C = R.clone() if epi == 3 else torch.empty(M, N, dtype=BF, device="cuda")
        for _ in range(3):
            gemm(C, epi, C if epi == 3 else None)
        e0.record()
        for _ in range(20):
            gemm(C, epi, C if epi == 3 else None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"policy {v} epi {epi}: fresh buffer: wrong elements seen by a GPU kernel {res[0][0]}, by D2H {res[0][1]};  pre-read buffer: GPU {res[1][0]}, D2H {res[1][1]};  "
              f"{us:8.1f} us/launch = {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  (of {M * N} elements)", flush=True)

# ---- the scoring pass's own dispatch (aigv_op_gemm_rows: row plans, fused tail slices), per GEMM of the four-clip step ----------------
import ctypes  # noqa: E402

SH = [("llm wqkv", [2176] * 4, 6144, 4096, 0), ("llm wo", [2176] * 4, 4096, 4096, 3), ("llm w1|w3", [2176] * 4, 28672, 4096, 4), ("llm w2", [2176] * 4, 4096, 14336, 3),
      ("vit qkv", [1025] * 32, 3072, 1024, 0), ("vit proj", [1025] * 32, 1024, 1024, 2), ("vit fc1", [1025] * 32, 4096, 1024, 1), ("vit fc2", [1025] * 32, 1024, 4096, 2)]
for name, lens, n, k, epi in SH:
    m = sum(lens)
    cu = [0]
    for x in lens:
        cu.append(cu[-1] + x)
    cu_a = (ctypes.c_int32 * len(cu))(*cu)
    a = (torch.randn(m, k, generator=g, device="cuda") * 0.5).to(BF)
    w = (torch.randn(n, k, generator=g, device="cuda") / math.sqrt(k)).to(BF)
    nout = n // 2 if epi == 4 else n
    bias = (torch.randn(n, generator=g, device="cuda") * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(n, generator=g, device="cuda") + 0.5).to(BF) if epi == 2 else None
    r0 = torch.randn(m, nout, generator=g, device="cuda").to(BF) if epi in (2, 3) else None
    outs, times = {}, {}
    for v in (0, 5, 6, 7):
        policy(v)
        c = r0.clone() if r0 is not None else torch.full((m, nout), 7.0, dtype=BF, device="cuda")

        def call(cc):
            native.check(lib.aigv_op_gemm_rows(ptr(a), k, ptr(w), k, ptr(cc), nout, ptr(bias), ptr(ls), ptr(cc) if r0 is not None else None, nout, cu_a, len(lens), n, k, epi, None))
        call(c)
        torch.cuda.synchronize()
        outs[v] = c.clone()
        cc = r0.clone() if r0 is not None else torch.empty(m, nout, dtype=BF, device="cuda")
        for _ in range(2):
            call(cc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call(cc)          # (the test entry point synchronises the stream per call: times include that)
        e1.record()
        torch.cuda.synchronize()
        times[v] = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name:10s} rows {m} N {n} K {k} epi {epi}: " + "  ".join(f"policy {v}: {times[v]:7.1f} us, differing elements vs plain {int((outs[v] != outs[0]).sum())}" for v in (0, 5, 6, 7)), flush=True)
