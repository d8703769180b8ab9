"""A few launches of the e4m3 form of the 256 kernel on the two dominant shapes (w1|w3 SwiGLU, w2 residual), for rocprofv3 --pmc passes."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    qa = torch.empty(M, K, dtype=torch.uint8, device='cuda'); sa = torch.empty(M, dtype=torch.float32, device='cuda')
    qw = torch.empty(N, K, dtype=torch.uint8, device='cuda'); sw = torch.empty(N, dtype=torch.float32, device='cuda')
    native.check(lib.aigv_op_quant_fp8_rows(ptr(A), K, M, K, ptr(qa), K, ptr(sa), None))
    native.check(lib.aigv_op_quant_fp8_rows(ptr(W), K, N, K, ptr(qw), K, ptr(sw), None))
    for _ in range(iters):
        native.check(lib.aigv_op_gemm_fp8(ptr(qa), K, ptr(qw), K, ptr(C), nout, ptr(sa), ptr(sw), None, None, ptr(resid), nout, M, N, K, epi, 0, None, None))
    torch.cuda.synchronize()
run(8704, 28672, 4096, 4)
run(8192, 4096, 14336, 3)
