"""Decode GEMVs (one x row, the 8B shapes) on the weight-streaming skinny kernel next to the vendor BLAS behind torch's F.linear: us per launch and TB/s of weights,
six rotating weight buffers per shape (no L2 / Infinity-Cache reuse between launches), plain store epilogue.  A yardstick, like scripts/gemm_vs_blas.py.
    python scripts/gemv_vs_blas.py"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16


def timed(call, n=120):
    for i in range(12): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): call(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in (("wqkv", 6144, 4096), ("wo", 4096, 4096), ("w2", 4096, 14336), ("w1|w3", 28672, 4096), ("lm-head", 92544, 4096)):
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(6)]
    x = torch.randn(1, K, device="cuda").to(BF)
    out = torch.empty(1, N, dtype=BF, device="cuda")
    ours = lambda i: native.check(lib.aigv_op_skinny_gemm(ptr(x), K, 1, ptr(Ws[i % 6]), K, N, K, None, None, N, ptr(out), N, 0, None))
    blas = lambda i: F.linear(x, Ws[i % 6])
    a = sorted(timed(ours) for _ in range(3))[1]
    b = sorted(timed(blas) for _ in range(3))[1]
    gb = N * K * 2
    torch.testing.assert_close(out.float(), F.linear(x, Ws[(12 + 120 - 1) % 6]).float(), rtol=3e-2, atol=3e-2)
    print(f"{name:8s} N={N:6d} K={K:6d}: skinny kernel {a:7.2f} us ({gb / a / 1e6:5.2f} TB/s)   vendor BLAS {b:7.2f} us ({gb / b / 1e6:5.2f} TB/s)   time ratio {a / b:5.3f}", flush=True)
