"""The prefill attention kernel next to torch's scaled_dot_product_attention (the flash kernel the ROCm build of torch ships) on the two headline shapes and on
the large regular one: same process, random bf16 data, chip warm.  A yardstick, like scripts/gemm_vs_blas.py.
    python scripts/attn_vs_sdpa.py"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr

lib = native.load()
BF = torch.bfloat16


def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(name, d, causal, h, hk, L, B):
    g = h // hk
    ld = hk * (g + 2) * d
    T = B * L
    qkv = (torch.randn(T, ld, device="cuda") * 0.5).to(BF)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    out = torch.empty(T, h * d, dtype=BF, device="cuda")
    base = qkv.data_ptr()
    pre = d ** -0.5 if not causal else 1.0
    post = 1.0 if not causal else math.sqrt(d)
    ours = lambda: native.check(lib.aigv_op_attention(base, ld, base + g * d * 2, ld, base + (g + 1) * d * 2, ld, ptr(out), h * d, ptr(cu), B, L, h, hk,
                                                      (g + 2) * d, (g + 2) * d, d, int(causal) | 2, post, pre, native.stream_ptr()))
    f = qkv.view(B, L, hk, g + 2, d)
    q = f[:, :, :, :g].reshape(B, L, h, d).transpose(1, 2).contiguous()
    k = f[:, :, :, g].transpose(1, 2).contiguous()
    v = f[:, :, :, g + 1].transpose(1, 2).contiguous()
    sdpa = lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=(g > 1))
    to = sorted(timed(ours) for _ in range(3))[1]
    try:
        ts = sorted(timed(sdpa) for _ in range(3))[1]
        ref = sdpa().transpose(1, 2).reshape(T, h * d)
        ours()
        torch.cuda.synchronize()
        err = (out.float() - ref.float()).abs().max().item()
    except Exception as e:      # a backend that does not take the shape
        ts, err = float("nan"), float("nan")
        print("   sdpa failed:", str(e)[:120])
    fl = 4.0 * B * L * L * d * h * (0.5 if causal else 1.0)
    print(f"{name:52s} this kernel {to:8.1f} us ({fl / to / 1e6:7.1f} TFLOP/s)   torch sdpa {ts:8.1f} us ({fl / ts / 1e6:7.1f} TFLOP/s)   time ratio {to / ts:5.3f}   max |diff| {err:.4f}", flush=True)


a = torch.randn(8192, 8192, device="cuda").to(BF)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(10): a @ a
    torch.cuda.synchronize()
case("InternViT: d 64, 16 heads, 32 x 1025, non-causal", 64, False, 16, 16, 1025, 32)
case("InternLM2: d 128, 32 / 8 heads, 4 x 2177, causal", 128, True, 32, 8, 2177, 4)
case("d 128, 64 / 8 heads, 4 x 2048, non-causal", 128, False, 64, 8, 2048, 4)
case("d 128, 32 / 8 heads, 4 x 8192, causal", 128, True, 32, 8, 8192, 4)
case("d 64, 16 heads, 32 x 1024, non-causal", 64, False, 16, 16, 1024, 32)
