import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def t(M, N, K, mode, epi=0, iters=20):
    A = torch.randn(M, K, device='cuda').to(BF); W = (torch.randn(N, K, device='cuda') / math.sqrt(K)).to(BF)
    nout = N // 2 if epi == 4 else N
    C = torch.empty(M, nout, dtype=BF, device='cuda')
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    call = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), nout, None, None, None, 0, None, 0, M, N, K, epi, None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"mode={mode} epi={epi} M={M:6d} N={N:6d} K={K:6d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s", flush=True)
for epi in (0, 4):
    for M in (8192, 8448, 8704, 8708, 8960):
        t(M, 28672, 4096, 2, epi)
    t(8708, 28672, 4096, 0, epi)
    t(4, 28672, 4096, 1, epi)
for M in (8192, 8704, 8708):
    t(M, 4096, 14336, 2, 3 if False else 0)
t(8708, 4096, 14336, 0)
t(516, 4096, 14336, 1)
