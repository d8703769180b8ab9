"""The scorer's GEMM shapes on this library's tile kernel next to the vendor BLAS behind torch (hipBLASLt / rocBLAS through F.linear), same process, random bf16
data, plain store epilogue, chip warm: a yardstick for "how far is the tile kernel from what the chip gives a tuned library GEMM on this shape".
    python scripts/gemm_vs_blas.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr

lib = native.load()
BF = torch.bfloat16
shapes = [("InternLM2 w1|w3", 8704, 28672, 4096), ("InternLM2 w2", 8704, 4096, 14336), ("InternLM2 wqkv", 8704, 6144, 4096), ("InternLM2 wo", 8704, 4096, 4096),
          ("InternViT fc1", 32768, 4096, 1024), ("InternViT qkv", 32768, 3072, 1024), ("InternViT proj", 32768, 1024, 1024), ("InternViT fc2", 32768, 1024, 4096),
          ("8192^3", 8192, 8192, 8192)]


def timed(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# warm the chip for ~2 s
a = torch.randn(8192, 8192, device="cuda").to(BF); b = torch.randn(8192, 8192, device="cuda").to(BF)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(10): F.linear(a, b)
    torch.cuda.synchronize()
print(f"{'shape':18s} {'M':>6s} {'N':>6s} {'K':>6s}   tile kernel us (TFLOP/s)    vendor BLAS us (TFLOP/s)   tile / BLAS time", flush=True)
for name, M, N, K in shapes:
    A = (torch.randn(M, K, device="cuda") * 0.5).to(BF)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    Cm = torch.empty(M, N, dtype=BF, device="cuda")
    native.check(lib.aigv_tune_gemm(2, 0.0))
    ours = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(Cm), N, None, None, None, N, None, 0, M, N, K, 0, native.stream_ptr()))
    blas = lambda: F.linear(A, W)
    res = []
    for rnd in range(3):
        res.append((timed(ours, 20), timed(blas, 20)))
    native.check(lib.aigv_tune_gemm(0, 0.0))
    o = sorted(r[0] for r in res)[1]; bl = sorted(r[1] for r in res)[1]
    fl = 2.0 * M * N * K
    print(f"{name:18s} {M:6d} {N:6d} {K:6d}   {o:9.1f} ({fl / o / 1e6:7.1f})        {bl:9.1f} ({fl / bl / 1e6:7.1f})        {o / bl:5.3f}", flush=True)
    torch.testing.assert_close(Cm.float(), F.linear(A, W).float(), rtol=2e-2, atol=2e-2)
