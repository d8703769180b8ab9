"""What would a weight prefetch into the Infinity Cache (MALL, 256 MB) / L2 buy a decode GEMV?  One x row, 8B shapes:
  cold: the weights rotate through > 600 MB of buffers (every launch streams from HBM, as in the decode step: 14.7 GB per token)
  warm: the same buffer every launch (resident in MALL / L2 as far as it fits) - the best case a perfect prefetcher could reach
python scripts/mall_prefetch_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
R = 1
for name, N, K, epi in (("wqkv", 6144, 4096, 0), ("wo", 4096, 4096, 1), ("w2", 4096, 14336, 1), ("w1|w3", 28672, 4096, 2)):
    nbuf = max(3, int(700e6 / (N * K * 2)) + 1)
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(nbuf)]
    x = torch.randn(R, K, device="cuda").to(BF)
    nout = N // 2 if epi == 2 else N
    res = torch.randn(R, nout, device="cuda").to(BF) if epi == 1 else None
    out = torch.empty(R, nout, dtype=BF, device="cuda")
    def call(i):
        native.check(lib.aigv_op_skinny_gemm(ptr(x), K, R, ptr(Ws[i]), K, N, K, None, ptr(res), nout, ptr(out), nout, epi, None))
    def run(rotate):
        for i in range(2 * nbuf): call(i % nbuf if rotate else 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 4 * nbuf
        e0.record()
        for i in range(n): call(i % nbuf if rotate else 0)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    c1, w1, c2, w2 = run(True), run(False), run(True), run(False)
    mb = N * K * 2 / 1e6
    print(f"{name:6s} {mb:6.1f} MB: cold {c1:6.2f} / {c2:6.2f} us ({mb / c1:5.2f} TB/s)   warm {w1:6.2f} / {w2:6.2f} us ({mb / w1:5.2f} TB/s)", flush=True)
    del Ws
