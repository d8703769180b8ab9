"""A few launches of the two dominant GEMM shapes (w1|w3 SwiGLU, w2 residual) through the dispatcher, for rocprofv3 --pmc passes."""
import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def run(M, N, K, epi, iters=6):
    A = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    W = (torch.randn(N, K, device='cuda') / math.sqrt(K)).to(BF)
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, device='cuda').to(BF) if epi == 3 else None
    C = torch.empty(M, nout, dtype=BF, device='cuda')
    for _ in range(iters):
        native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), nout, None, None, ptr(resid), nout, None, 0, M, N, K, epi, None))
    torch.cuda.synchronize()
run(8704, 28672, 4096, 4)
run(8192, 4096, 14336, 3)
