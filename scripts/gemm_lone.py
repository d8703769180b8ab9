import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def t(M, N, K, mode, iters=20):
    A = torch.randn(M, K, device='cuda').to(BF); W = (torch.randn(N, K, device='cuda') / math.sqrt(K)).to(BF)
    C = torch.empty(M, N, dtype=BF, device='cuda')
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    call = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), N, None, None, None, 0, None, 0, M, N, K, 0, None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tiles = ((M + 127) // 128) * (N // 128) if mode == 1 else ((M + 255) // 256) * (N // 256)
    print(f"mode={mode} M={M:6d} N={N:6d} K={K:6d} tiles={tiles:5d}: {us:8.1f} us", flush=True)
for K in (1024, 4096, 14336):
    for (M, N) in [(128, 128), (4, 28672), (128, 4096), (128, 8192 * 4), (256, 8192 * 4), (512, 8192 * 4), (516, 4096), (516, 6144), (1024, 8192*4), (2048, 8192 * 4)]:
        t(M, N, K, 1)
    for (M, N) in [(256, 256), (256, 8192*8), (512, 8192 * 8), (768, 8192*8), (1024, 8192*8)]:
        t(M, N, K, 2)
