"""A few launches of the two headline attention shapes, for rocprofv3 --pmc passes (scripts/pmc_sq_summary.py)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
import os
lib = native.load()
native.check(lib.aigv_tune_attention(int(os.environ.get("ATTN_KERNEL", "0"))))   # 0 default, 4 / 8 waves per workgroup
BF = torch.bfloat16
def run(d, causal, h, hk, lens, iters=6):
    T = sum(lens); g = h // hk
    ld = hk * (g + 2) * d
    qkv = torch.randn(T, ld, device='cuda').to(BF)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device='cuda')
    out = torch.empty(T, h * d, dtype=BF, device='cuda')
    base = qkv.data_ptr()
    pre = d ** -0.5 if not causal else 1.0
    post = 1.0 if not causal else math.sqrt(d)
    for _ in range(iters):
        native.check(lib.aigv_op_attention(base, ld, base + g * d * 2, ld, base + (g + 1) * d * 2, ld, ptr(out), h * d, ptr(cu), len(lens), max(lens), h, hk, (g + 2) * d, (g + 2) * d, d, int(causal), post, pre, None))
    torch.cuda.synchronize()
run(64, False, 16, 16, [1024] * 32)
run(128, True, 32, 8, [2176] * 4)
