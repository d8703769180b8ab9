"""Decode GEMVs (one x row) on the 8B shapes: us per launch and TB/s of weights.  A/B of the 8-wave form:
python scripts/decode_gemv_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name, N, K, epi in (("wqkv", 6144, 4096, 0), ("wo", 4096, 4096, 1), ("w2", 4096, 14336, 1), ("w1|w3", 28672, 4096, 2)):
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(6)]          # rotate buffers: no L2 / MALL reuse between launches
    x = torch.randn(R, K, device="cuda").to(BF)
    nout = N // 2 if epi == 2 else N
    res = torch.randn(R, nout, device="cuda").to(BF) if epi == 1 else None
    out = torch.empty(R, nout, dtype=BF, device="cuda")
    def call(i):
        native.check(lib.aigv_op_skinny_gemm(ptr(x), K, R, ptr(Ws[i % 6]), K, N, K, None, ptr(res), nout, ptr(out), nout, epi, None))
    for i in range(12): call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 120
    e0.record()
    for i in range(n): call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:6s} N={N:6d} K={K:6d} R={R}: {us:7.2f} us  {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
