"""fp8 (e4m3) operand form of the 256x256 kernel beside the bf16 one, same shapes, same process, interleaved:
python scripts/gemm_fp8_bench.py.  Rows are padded to whole 256-row rounds (the fp8 form has no remainder-band planner yet);
plain store epilogue on both sides.  The quantisation kernel is timed separately (it would be fused into the producer)."""
import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
SHAPES = [("llm_w13", 8704, 28672, 4096), ("llm_w2", 8704, 4096, 14336), ("llm_wo", 8704, 4096, 4096), ("llm_wqkv", 8704, 6144, 4096),
          ("vit_fc1", 32768, 4096, 1024), ("vit_fc2", 32768, 1024, 4096), ("sq8k", 8192, 8192, 8192)]
def timed(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for rep in range(2):
    for name, M, N, K in SHAPES:
        g = torch.Generator(device='cuda').manual_seed(1)
        A = (torch.randn(M, K, generator=g, device='cuda') * 0.5).to(BF)
        W = (torch.randn(N, K, generator=g, device='cuda') / math.sqrt(K)).to(BF)
        C = torch.empty(M, N, dtype=BF, device='cuda'); C8 = torch.empty_like(C)
        qa = torch.empty(M, K, dtype=torch.uint8, device='cuda'); sa = torch.empty(M, dtype=torch.float32, device='cuda')
        qw = torch.empty(N, K, dtype=torch.uint8, device='cuda'); sw = torch.empty(N, dtype=torch.float32, device='cuda')
        native.check(lib.aigv_op_quant_fp8_rows(ptr(W), K, N, K, ptr(qw), K, ptr(sw), None))
        quant = lambda: native.check(lib.aigv_op_quant_fp8_rows(ptr(A), K, M, K, ptr(qa), K, ptr(sa), None))
        bf = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), N, None, None, None, N, None, 0, M, N, K, 0, None))
        f8 = lambda: native.check(lib.aigv_op_gemm_fp8(ptr(qa), K, ptr(qw), K, ptr(C8), N, ptr(sa), ptr(sw), None, None, None, 0, M, N, K, 0, 0, None, None))
        quant(); t_bf = timed(bf); t_f8 = timed(f8); t_q = timed(quant)
        rel = ((C8.float() - C.float()).norm() / C.float().norm()).item()
        fl = 2 * M * N * K / 1e9
        print(f"{name:9s} M={M:6d} N={N:6d} K={K:6d}: bf16 {t_bf*1e3:8.1f} us {fl/t_bf:7.1f} TF/s | fp8 {t_f8*1e3:8.1f} us {fl/t_f8:7.1f} TF/s "
              f"({t_bf/t_f8:4.2f}x) | quant A {t_q*1e3:6.1f} us ({M*K*3/t_q/1e6:6.0f} GB/s) | rel diff fp8 vs bf16 {rel:.4f}", flush=True)
