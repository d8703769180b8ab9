"""GEMM microbenchmark + correctness on the scorer's shapes: python scripts/gemm_bench.py [mode ...]"""
import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
SHAPES = [  # (name, M, N, K, epi)
    ("llm_w13", 8708, 28672, 4096, 4), ("llm_w2", 8708, 4096, 14336, 3), ("llm_wo", 8708, 4096, 4096, 3),
    ("llm_wqkv", 8708, 6144, 4096, 0), ("vit_qkv", 32800, 3072, 1024, 0), ("vit_proj", 32800, 1024, 1024, 2),
    ("vit_fc1", 32800, 4096, 1024, 1), ("vit_fc2", 32800, 1024, 4096, 2), ("proj1", 8192, 4096, 4096, 1),
    ("sq4k", 4096, 4096, 4096, 0), ("sq8k", 8192, 8192, 8192, 0),
]
def run(name, M, N, K, epi, mode, check=True, iters=10):
    g = torch.Generator(device='cuda').manual_seed(1)
    A = (torch.randn(M, K, generator=g, device='cuda') * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g, device='cuda') / math.sqrt(K)).to(BF)
    nout = N // 2 if epi == 4 else N
    bias = (torch.randn(N, generator=g, device='cuda') * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g, device='cuda') + 0.5).to(BF) if epi == 2 else None
    resid = torch.randn(M, nout, generator=g, device='cuda').to(BF) if epi in (2, 3) else None
    C = torch.empty(M, nout, dtype=BF, device='cuda')
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    def call():
        native.check(lib.aigv_op_gemm(ptr(A), K, ptr(W), K, ptr(C), nout, ptr(bias), ptr(ls), ptr(resid), nout, None, 0, M, N, K, epi, None))
    call(); torch.cuda.synchronize()
    err = ""
    if check:
        acc = A.float() @ W.float().t()
        rb = lambda t: t.to(BF).float()
        if epi == 4:
            blk = acc.view(M, N // 32, 2, 16)
            gt, up = rb(blk[:, :, 0]).reshape(M, -1), rb(blk[:, :, 1]).reshape(M, -1)
            want = rb(rb(torch.nn.functional.silu(gt)) * up)
        else:
            y = rb(acc + bias.float()) if bias is not None else rb(acc)
            if epi == 1: y = rb(torch.nn.functional.gelu(y))
            if epi == 2: y = rb(y * ls.float())
            if epi in (2, 3): y = rb(resid.float() + y)
            want = y
        d = (C.float() - want).abs()
        tol = 2.0 ** -6 * want.abs() + 2.0 ** -7 * want.abs().max() * (1 if epi in (2, 3) else 0.01)
        nbad = int((d > tol).sum())
        err = f"bad={nbad} maxerr={d.max().item():.3g} nan={int(torch.isnan(C.float()).sum())}"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): call()
    e0.record()
    for _ in range(iters): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:9s} M={M:6d} N={N:6d} K={K:6d} epi={epi} mode={mode}: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {err}", flush=True)
modes = [int(x) for x in sys.argv[1:]] or [1, 2, 0]
for rep in range(2):          # interleaved rounds in ONE process (variance between runs/devices is ~10 %)
    for sh in SHAPES:
        for mode in modes:
            run(*sh, mode, check=(rep == 0 and sh[1] * sh[2] <= 8708 * 28672))
