import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def run(M, N, K, epi, mode, iters=30):
    g = torch.Generator(device='cuda').manual_seed(1)
    A = (torch.randn(M, K, generator=g, device='cuda') * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g, device='cuda') / math.sqrt(K)).to(BF)
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g, device='cuda').to(BF) if epi in (2, 3) else None
    C = torch.empty(M, nout, dtype=BF, device='cuda')
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    call = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), nout, None, None, ptr(resid), nout, None, 0, M, N, K, epi, None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rep in range(2):
  for name, M, N, K, epi in (("wo body 1 clip", 2048, 4096, 4096, 3), ("w2 body 1 clip", 2048, 4096, 14336, 3), ("wqkv body 1 clip", 2048, 6144, 4096, 0), ("vit fc2 body 8 frames", 8192, 1024, 4096, 3), ("vit proj body 8 frames", 8192, 1024, 1024, 3),
                             ("wo body 2 clips", 4096, 4096, 4096, 3), ("w2 body 2 clips", 4096, 4096, 14336, 3)):
    print(f"{name:24s} M={M} N={N} K={K}: 256x256 {run(M,N,K,epi,2+32):7.1f} us   co 2/CU form {run(M,N,K,epi,4+32):7.1f} us   LONE form {run(M,N,K,epi,4+16*7):7.1f} us", flush=True)
native.check(lib.aigv_tune_gemm(0+32, 0.0))
