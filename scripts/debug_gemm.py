import math, sys, torch
sys.path.insert(0, '.')
from aigv_assessor_amd import native
from aigv_assessor_amd.native import ptr
lib = native.load()
BF = torch.bfloat16
def run(M, N, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    want = (A.float() @ W.float().t()).to(BF).float()
    dA, dW = A.cuda(), W.cuda()
    dC = torch.full((M, N), float('nan'), dtype=BF, device='cuda')
    rc = lib.aigv_op_gemm(ptr(dA), K, ptr(dW), K, ptr(dC), N, None, None, None, 0, None, 0, M, N, K, 0, None)
    torch.cuda.synchronize()
    got = dC.float().cpu()
    bad = (got - want).abs() > 0.02 * want.abs() + 1e-3
    print(f"M={M} N={N} K={K} rc={rc} bad={int(bad.sum())}/{bad.numel()} nan={int(torch.isnan(got).sum())}")
    if bad.any():
        tm = (M + 127) // 128; tn = N // 128
        for i in range(tm):
            row = []
            for j in range(tn):
                row.append(int(bad[i*128:(i+1)*128, j*128:(j+1)*128].sum()))
            if any(row): print("  tile row", i, row)
        r, c = bad.nonzero()[0].tolist()
        print("  first bad", r, c, got[r, c].item(), want[r, c].item())
        # which k-slices are missing? compare with partial sums
        for ks in range(K // 64):
            part = (A[r].float()[ks*64:(ks+1)*64] * W[c].float()[ks*64:(ks+1)*64]).sum().item()
            print("   kslice", ks, part)
for shp in [(128,128,64),(128,128,128),(128,128,192),(128,128,256),(256,128,192),(128,256,192),(128,384,192),(2050,384,192),(2050,384,128),(2050,256,192),(300,256,128),(1024,1024,1024)]:
    run(*shp)
