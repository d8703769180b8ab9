"""Attention microbenchmark on the headline shapes: python scripts/attn_bench.py"""
import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def run(name, d, causal, h, hk, lens, iters=60, uniform=True):
    T = sum(lens); g = h // hk
    ld = hk * (g + 2) * d
    qkv = torch.randn(T, ld, device='cuda').to(BF)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device='cuda')
    out = torch.empty(T, h * d, dtype=BF, device='cuda')
    base = qkv.data_ptr()
    pre = d ** -0.5 if not causal else 1.0
    post = 1.0 if not causal else math.sqrt(d)
    call = lambda: native.check(lib.aigv_op_attention(base, ld, base + g * d * 2, ld, base + (g + 1) * d * 2, ld, ptr(out), h * d, ptr(cu), len(lens), max(lens), h, hk, (g + 2) * d, (g + 2) * d, d, int(causal) | (2 if uniform else 0), post, pre, None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = sum(4.0 * (L * (L + 1) / 2 if causal else L * L) * d * h for L in lens)
    print(f"{name}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s", flush=True)
import os
for rep in range(2):
  for nw in ((int(os.environ["ATTN_ONLY"]),) if os.environ.get("ATTN_ONLY") else (0,) if not os.environ.get("AB_WAVES") else (4, 8, 0)):
    native.check(lib.aigv_tune_attention(nw))
    print(f"-- kernel choice: {({0: 'default', 4: '4 waves x 32 rows', 8: '8 waves'})[nw]}")
    run("vit  d64  32x1025 h16", 64, False, 16, 16, [1025] * 32)
    run("vit  d64  32x1024 h16", 64, False, 16, 16, [1024] * 32)
    run("llm  d128 4x2177 h32/8", 128, True, 32, 8, [2177] * 4)
    run("llm  d128 1x4281 h32/8", 128, True, 32, 8, [4281])
    if os.environ.get("AB_WAVES") or os.environ.get("ATTN_ONLY"):
        run("d128 noncausal 8x1024 h16", 128, False, 16, 16, [1024] * 8)
        run("llm  d128 4x2176 h32/8", 128, True, 32, 8, [2176] * 4)
native.check(lib.aigv_tune_attention(0))
