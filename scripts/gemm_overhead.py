"""Per-tile fixed overhead of the 256x256 GEMM kernel: time vs K at fixed M, N (slope = K-tile time, intercept = prologue + epilogue + dispatch)."""
import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def run(M, N, K, epi, mode, iters=20):
    A = (torch.randn(M, K, device='cuda') * 0.5).to(BF)
    W = (torch.randn(N, K, device='cuda') / math.sqrt(K)).to(BF)
    nout = N // 2 if epi == 4 else N
    bias = torch.zeros(N, device='cuda').to(BF) if epi in (0, 1, 2) else None
    ls = torch.ones(N, device='cuda').to(BF) if epi == 2 else None
    resid = torch.randn(M, nout, device='cuda').to(BF) if epi in (2, 3) else None
    C = torch.empty(M, nout, dtype=BF, device='cuda')
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    call = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), nout, ptr(bias), ptr(ls), ptr(resid), nout, None, 0, M, N, K, epi, None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
modes = [int(x) for x in sys.argv[1:]] or [2]
for mode in modes:
    for (M, N) in ((32768, 4096),):
        rounds = (M // 256) * (N // 256) / 256
        for epi in (0, 1, 3, 4):
            ts = {K: run(M, N, K, epi, mode) for K in (64, 256, 1024, 4096)}
            slope = (ts[4096] - ts[1024]) / (3072 / 64) / rounds
            print(f"mode={mode} M={M} N={N} epi={epi} rounds={rounds:.0f}: " + " ".join(f"K={k}:{v:7.1f}us" for k, v in ts.items()) +
                  f" | per-round: ktile={slope:.3f}us fixed(K=64)={(ts[64]) / rounds - slope:.2f}us fixed(fit@1024)={(ts[1024] / rounds - 16 * slope):.2f}us", flush=True)
