"""The prefill attention kernel on its two headline shapes and on the large regular shape kernel guides quote (non-causal and causal,
64 heads / 8 kv heads, N = 2048, d = 128, batch 4 and 16): isolated TFLOP/s by HIP events on random data, after a 1 s warm-up.
Separates what the kernel's structure yields from what the headline shapes cost (short causal sequences: 17 query blocks, ragged diagonals).
    python scripts/attn_shapes.py"""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16

def run(name, d, causal, h, hk, lens, secs=1.0):
    T = sum(lens); g = h // hk
    ld = hk * (g + 2) * d
    qkv = (torch.randn(T, ld, device='cuda') * 0.5).to(BF)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device='cuda')
    out = torch.empty(T, h * d, dtype=BF, device='cuda')
    base = qkv.data_ptr()
    pre = d ** -0.5 if not causal else 1.0
    post = 1.0 if not causal else math.sqrt(d)
    call = lambda: native.check(lib.aigv_op_attention(base, ld, base + g * d * 2, ld, base + (g + 1) * d * 2, ld, ptr(out), h * d, ptr(cu), len(lens), max(lens), h, hk,
                                                      (g + 2) * d, (g + 2) * d, d, int(causal), post, pre, native.stream_ptr()))
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(20): call()
        torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = sum(4.0 * L * L * d * h * (0.5 if causal else 1.0) for L in lens)
    print(f"{name:58s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s ({flops / us / 1e6 / 2500:.3f} of 2.5 PF)", flush=True)

run("InternViT: d 64, 16 heads, 32 x 1025, non-causal", 64, False, 16, 16, [1025] * 32)
run("InternLM2: d 128, 32/8 heads, 4 x 2177, causal", 128, True, 32, 8, [2177] * 4)
run("d 128, 32/8 heads, 4 x 2176, causal", 128, True, 32, 8, [2176] * 4)
run("d 128, 64/8 heads, 4 x 2048, non-causal", 128, False, 64, 8, [2048] * 4)
run("d 128, 64/8 heads, 16 x 2048, non-causal", 128, False, 64, 8, [2048] * 16)
run("d 128, 64/8 heads, 16 x 2048, causal", 128, True, 64, 8, [2048] * 16)
run("d 128, 32/8 heads, 4 x 8192, causal", 128, True, 32, 8, [8192] * 4)
run("d 64, 16 heads, 32 x 1024, non-causal", 64, False, 16, 16, [1024] * 32)
run("d 64, 16 heads, 8 x 4096, non-causal", 64, False, 16, 16, [4096] * 8)
native.check(lib.aigv_tune_default(8, 1))       # AIGV_TUNE_ATTN_LEAD_KEY: key 0 merged in the epilogue, full tiles over keys 1..
run("InternViT 32 x 1025 with the lead-key form (opt-in)", 64, False, 16, 16, [1025] * 32)
native.check(lib.aigv_tune_default(8, 0))
