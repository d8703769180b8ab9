import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (name, N, K, epi) in (("wo", 4096, 4096, 3), ("wqkv", 6144, 4096, 0), ("w2", 4096, 14336, 3)):
    M = 8704
    A = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    W = (torch.randn(N, K, device='cuda') / math.sqrt(K)).to(BF)
    R = torch.randn(M, N, device='cuda').to(BF) if epi == 3 else None
    C = torch.empty(M, N, dtype=BF, device='cuda')
    ws = torch.empty(8 * 1024 * N, dtype=torch.float32, device='cuda')
    full = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), N, None, None, ptr(R), N, None, 0, M, N, K, epi, None))
    top = lambda rows: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), N, None, None, ptr(R), N, None, 0, rows, N, K, epi, None))
    def mid(rows0, rows, S):
        a = A[rows0:]; c = C[rows0:]; r = R[rows0:] if R is not None else None
        native.check(lib.aigv_op_gemm_splitk256(ptr(a), K, ptr(W), K, ptr(c), N, None, None, ptr(r), N, rows, N, K, epi, S, ptr(ws), None))
    def rem128(rows0, rows, S):
        a = A[rows0:]; c = C[rows0:]; r = R[rows0:] if R is not None else None
        if S > 1: native.check(lib.aigv_op_gemm_splitk(ptr(a), K, ptr(W), K, ptr(c), N, None, None, ptr(r), N, rows, N, K, epi, S, ptr(ws), None))
        else:
            native.check(lib.aigv_tune_gemm(1, 0.0)); native.check(lib.aigv_op_gemm(ptr(a), K, ptr(W), K, ptr(c), N, None, None, ptr(r), N, None, 0, rows, N, K, epi, None)); native.check(lib.aigv_tune_gemm(0, 0.0))
    p = (torch.zeros(7, dtype=torch.int32)); import ctypes
    pl = (ctypes.c_int * 7)(); est = ctypes.c_double(); native.check(lib.aigv_plan_gemm(M, N, K, epi, pl, ctypes.byref(est)))
    print(f"{name}: planner {list(pl)} est {est.value:.0f} us; measured full dispatch {t(full):.1f} us")
    print(f"   top 8192 rows: {t(lambda: top(8192)):.1f} us | top 7936 rows: {t(lambda: top(7936)):.1f} us")
    for S in (2, 4, 8):
        if (K // 64) % S == 0: print(f"   mid 512 rows split-K {S} (256 kernel): {t(lambda: mid(8192, 512, S)):.1f} us")
    for S in (1, 2, 4):
        print(f"   512 rows on the 128 kernel, S={S}: {t(lambda: rem128(8192, 512, S)):.1f} us   | 768 rows: {t(lambda: rem128(7936, 768, S)):.1f} us")
