"""HBM write / read / copy rates seen by simple torch kernels (context for the GEMM epilogue store bursts)."""
import torch
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (32, 256, 1024, 4096):
    x = torch.empty(mb * 2**20 // 2, dtype=torch.bfloat16, device='cuda'); y = torch.empty_like(x)
    w = t(lambda: x.zero_()); c = t(lambda: y.copy_(x)); r = t(lambda: x.view(torch.int32).sum())
    print(f"{mb:5d} MB: write {mb/1024/w/1e3*1.0737:6.2f} TB/s  copy(r+w) {2*mb/1024/c/1e3*1.0737:6.2f} TB/s  read {mb/1024/r/1e3*1.0737:6.2f} TB/s")
