import sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
V, H = 92553, 4096
W = (torch.randn(V, H, device='cuda') * 0.02).to(BF)
for R in (4, 10, 40, 44, 64):
    h = torch.randn(R, H, device='cuda').to(BF)
    packed = torch.zeros(64, dtype=torch.int64, device='cuda'); idx = torch.empty(R, dtype=torch.int64, device='cuda')
    call = lambda: native.check(lib.aigv_op_lm_head_argmax(ptr(h), R, H, ptr(W), V, ptr(packed), ptr(idx), None, None))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = (h.float() @ W.float().t()).to(BF).float().argmax(-1)
    print(f"R={R}: {us:7.1f} us  {V*H*2/us/1e6:5.2f} TB/s of weights  argmax agrees: {int((ref == idx).sum())}/{R}")
