"""Per-GEMM A/B of the row-plan dispatch on the scorer's shapes, one clip and four: today's dispatch (256x256 body + split-K tail + finalize),
everything in ONE launch of the co-resident 256x128 kernel (full K), everything in one launch of the 256x256 kernel (full K).

    python scripts/gemm_rows_ab.py"""
import ctypes, math, sys
sys.path.insert(0, ".")
import torch
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
LLM = [("wqkv", 6144, 4096, 0), ("wo", 4096, 4096, 3), ("w1|w3", 28672, 4096, 4), ("w2", 4096, 14336, 3)]
VIT = [("vit qkv", 3072, 1024, 0), ("vit proj", 1024, 1024, 2), ("vit fc1", 4096, 1024, 1), ("vit fc2", 1024, 4096, 2)]
ARMS = [("rows: 256 body + split-K tails", 0 + 32, 0), ("one launch, co-resident 256x128", 4 + 32, 0), ("one launch, 256x128 LONE form", 4 + 16 * 7, 0), ("one launch, 256x256", 2 + 32, 0)]


def run(lens, N, K, epi, mode, co_kmax, iters=30):
    M = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g, device="cuda") / math.sqrt(K)).to(BF)
    nout = N // 2 if epi == 4 else N
    bias = (torch.randn(N, generator=g, device="cuda") * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g, device="cuda") + 0.5).to(BF) if epi == 2 else None
    resid = torch.randn(M, nout, generator=g, device="cuda").to(BF) if epi in (2, 3) else None
    C = torch.empty(M, nout, dtype=BF, device="cuda")
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    cua = (ctypes.c_int32 * len(cu))(*cu)
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    native.check(lib.aigv_tune_co_gemm(co_kmax))
    # (aigv_op_gemm_rows synchronises per call: time the launches with events inside a batch of calls on the stream - the sync is outside the events)
    call = lambda: native.check(lib.aigv_op_gemm_rows(ptr(A), K, ptr(W), K, ptr(C), nout, ptr(bias), ptr(ls), ptr(resid), nout, cua, len(lens), N, K, epi, None))
    for _ in range(3):
        call()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for label, shapes, lens_list in (("InternLM2", LLM, ([2176], [2176] * 4)), ("InternViT", VIT, ([1025] * 8, [1025] * 32))):
    for lens in lens_list:
        for name, N, K, epi in shapes:
            out = []
            for arm, mode, ck in ARMS:
                out.append(f"{run(lens, N, K, epi, mode, ck):8.1f}")
            print(f"{label} {name:9s} {len(lens):2d} x {lens[0]:4d} rows  N={N:5d} K={K:5d}:  " + "  ".join(f"{a[0]}: {o} us" for a, o in zip(ARMS, out)), flush=True)
native.check(lib.aigv_tune_gemm(0 + 32, 0.0)); native.check(lib.aigv_tune_co_gemm(1024))
