"""What bounds the co-resident 256x128 GEMM kernel's K loop: a diagnostic build of gemmco.hip (-DAIGV_CO_DIAG) with two extra variants whose
RESULTS ARE GARBAGE - no LDS-DMA requests / vmcnt waits inside the K loop (what the loop costs without its memory stream) and, on top, no
workgroup barriers (without its rendezvous) - timed next to the shipped schedule and the 256x256 kernel on the scorer's shapes.

    python scripts/gemmco_diag.py build      # build container: scripts/_abl/libaigv_codiag.so
    python scripts/gemmco_diag.py run        # MI355X

The product build never defines the macro."""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aigv-assessor_amd")
OUT = os.path.join(ROOT, "scripts", "_abl")
LIB = os.path.join(OUT, "libaigv_codiag.so")

if sys.argv[1:2] == ["build"]:
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("aigv_assessor_amd.build")
    b.build()
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES if s != "gemmco.hip"]
    o = os.path.join(OUT, "gemmco_diag.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + b.FLAGS + ["-DAIGV_CO_DIAG"] + sys.argv[2:] + ["-c", os.path.join(PKG, "csrc", "gemmco.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [o])
    os.remove(o)
    print("built", LIB)
else:
    os.environ["AIGV_AMD_LIB"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    from aigv_assessor_amd import native
    from aigv_assessor_amd.native import ptr
    lib = native.load()
    BF = torch.bfloat16
    SHAPES = [("vit_qkv", 32768, 3072, 1024, 0), ("vit_proj", 32768, 1024, 1024, 2), ("vit_fc1", 32768, 4096, 1024, 1), ("vit_fc2", 32768, 1024, 4096, 2),
              ("llm_wo", 8704, 4096, 4096, 3), ("sq8k", 8192, 8192, 8192, 0)]
    # aigv_tune_gemm mode words: kernel + 16 * (1 + variant); variant 1 = the shipped schedule of either kernel
    MODES = [("256x256 kernel", 2 + 32), ("co-resident, shipped (interleaved)", 4 + 32), ("co-resident, block schedule", 4 + 16),
             ("co-resident, NO DMA in the K loop [garbage]", 4 + 48), ("co-resident, NO DMA, NO barriers [garbage]", 4 + 64),
             ("co-resident, DMA but NO vmcnt waits [garbage]", 4 + 80), ("co-resident, A-unit DMA only, NO waits [garbage]", 4 + 96)]

    def run(M, N, K, epi, mode, iters=20):
        g = torch.Generator(device="cuda").manual_seed(1)
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(BF)
        W = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(BF)
        nout = N // 2 if epi == 4 else N
        bias = (torch.randn(N, generator=g, device="cuda") * 0.1).to(BF) if epi in (0, 1, 2) else None
        ls = (torch.rand(N, generator=g, device="cuda") + 0.5).to(BF) if epi == 2 else None
        resid = torch.randn(M, nout, generator=g, device="cuda").to(BF) if epi in (2, 3) else None
        C = torch.empty(M, nout, dtype=BF, device="cuda")
        native.check(lib.aigv_tune_gemm(mode, 0.0))
        call = lambda: native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), nout, ptr(bias), ptr(ls), ptr(resid), nout, None, 0, M, N, K, epi, None))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    for rep in range(2):
        for name, M, N, K, epi in SHAPES:
            for label, mode in MODES:
                us = run(M, N, K, epi, mode)
                print(f"{name:9s} M={M} N={N} K={K} epi={epi}  {label:48s} {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
