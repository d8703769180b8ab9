"""Does the number of workgroups per CU matter for a decode GEMV?  SwiGLU and store forms (K = 4096, one x row) at widths
that give 2.5 / 3 / 3.5 / 4 workgroups per CU (256 CUs): us per launch and TB/s of weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
K, R = 4096, 1
for epi, per in ((2, 32), (0, 16)):
    for wgs in (192, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 1792, 2048):
        N = wgs * per
        Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(6)]
        x = torch.randn(R, K, device="cuda").to(BF)
        nout = N // 2 if epi == 2 else N
        out = torch.empty(R, nout, dtype=BF, device="cuda")
        def call(i):
            native.check(lib.aigv_op_skinny_gemm(ptr(x), K, R, ptr(Ws[i % 6]), K, N, K, None, None, nout, ptr(out), nout, epi, None))
        for i in range(12): call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 120
        e0.record()
        for i in range(n): call(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"epi={epi} workgroups={wgs:5d} ({wgs / 256:.2f} per CU) N={N:6d}: {us:7.2f} us  {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
        del Ws
