"""Per-kernel parity tests (MI355X only): every op is called through the C ABI (libaigv_amd.so) and compared
with a plain PyTorch fp32 reference of the same op that carries the reference's bf16 rounding points.

Tolerances (written per test): GEMM-class outputs must equal the rounded fp32 reference up to 1 bf16 ulp
(2^-8 relative) on a small fraction of elements — the only legitimate difference is fp32 summation order
inside the MFMA chain.  Attention is compared against fp64 truth next to the reference's eager bf16 path:
the kernel must be at least as accurate as that path (flash-style accumulation cannot match it bitwise).
Byte-moving ops (im2col, pixel-shuffle) are bit-exact.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16

# The families below that re-prove ``torch.equal`` between two kernel forms of the SAME arithmetic (tile kernels, tile orders, row plans,
# fused / separate tails) run, by default, every shape ONCE and every epilogue at least once (the diagonal of shapes x epilogues); the full
# cross product (5 epilogues x 4-9 shapes x up to 5 modes: ~190 more cases) runs under AIGV_FULL_MATRIX=1 - the driver's `-m gpu` step has
# 900 s for the whole suite (tests/manual/README.md).  The numerics tests against the fp32 reference are not thinned.
FULL_MATRIX = os.environ.get("AIGV_FULL_MATRIX") == "1"


def per_epilogue(shapes, second=(0, 1, 2, 3, 4)):
    """``shapes`` x ``second`` (epilogues, or kernel modes) as parametrize cases: the cross product under AIGV_FULL_MATRIX=1, else its diagonal."""
    if FULL_MATRIX:
        return [tuple(sh) + (e,) for sh in shapes for e in second]
    n = max(len(shapes), len(second))
    return [tuple(shapes[i % len(shapes)]) + (second[i % len(second)],) for i in range(n)]


@pytest.fixture(scope="module")
def lib():
    from aigv_assessor_amd import native
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return native.load()


_KEEP = []


@pytest.fixture(autouse=True)
def _release_device_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def dev(t):
    """Upload and KEEP a reference until the test ends: kernels run asynchronously on raw pointers."""
    d = t.cuda().contiguous()
    _KEEP.append(d)
    return d


def rb(t):  # one bf16 rounding point
    return t.to(BF).float()


def ulp_check(got, want, frac=0.02, max_ulps=2, atol_rel=2e-5):
    """got/want: bf16-valued float tensors.  Differences only at the 1-ulp level, on a small fraction.
    fp32 summation-order noise is absolute (~1e-6 of the operand scale), so results that cancel to ~0 get an
    absolute allowance next to the relative (ulp) one."""
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    ulp = (want.abs().clamp_min(1e-30)).log2().floor().exp2() * 2.0 ** -7
    err = (got - want).abs()
    atol = atol_rel * float(want.abs().max())
    nbad = (err > atol).float().mean().item()
    worst = ((err - atol).clamp_min(0) / ulp).max().item()
    assert worst <= max_ulps + 1e-3, f"worst error {worst:.2f} ulp"
    assert nbad <= frac, f"{nbad:.4f} of elements differ"


def sync(rc, lib_):
    from aigv_assessor_amd import native
    native.check(rc)
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------
def gemm_ref(A, W, epi, bias=None, ls=None, resid=None, pos=None, np_=0):
    acc = A.float() @ W.float().t()
    if epi == 4:
        n = W.shape[0]
        blk = acc.view(acc.shape[0], n // 32, 2, 16)
        g, u = rb(blk[:, :, 0, :]).reshape(acc.shape[0], -1), rb(blk[:, :, 1, :]).reshape(acc.shape[0], -1)
        return rb(rb(torch.nn.functional.silu(g)) * u)
    if bias is not None:
        acc = acc + bias.float()
    y = rb(acc)
    if epi == 1:
        y = rb(torch.nn.functional.gelu(y))
    if epi == 2:
        y = rb(y * ls.float())
    if epi in (2, 3):
        y = rb(resid.float() + y)
    if epi == 5:
        m = torch.arange(A.shape[0])
        y = rb(y + pos.float()[(m % np_) + 1])
    return y


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (2050, 384, 192), (128, 128, 64), (1, 128, 64), (4099, 1024, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm_epilogues(lib, M, N, K, epi):
    g = torch.Generator().manual_seed(M * 7 + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) * (1.0 / math.sqrt(K))).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g) + 0.5).to(BF) if epi == 2 else None
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g).to(BF) if epi in (2, 3) else None
    want = gemm_ref(A, W, epi, bias, ls, resid)
    dA, dW = dev(A), dev(W)
    dC = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
    db, dl, dr = (dev(t) if t is not None else None for t in (bias, ls, resid))
    from aigv_assessor_amd.native import ptr
    sync(lib.aigv_op_gemm(ptr(dA), K, ptr(dW), K, ptr(dC), nout, ptr(db), ptr(dl), ptr(dr), nout, None, 0, M, N, K, epi,
                          None), lib)
    # a residual add can cancel: the 1-ulp difference of the Linear output then dwarfs |want| -> absolute allowance
    ulp_check(dC, want, frac=0.03 if epi in (1, 4) else 0.02, max_ulps=4 if epi in (1, 4) else 2,
              atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)


def _gemm_rows_case(g, lens, N, K, epi):
    M = sum(lens)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) * (1.0 / math.sqrt(K))).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g) + 0.5).to(BF) if epi == 2 else None
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g).to(BF) if epi in (2, 3) else None
    return A, W, bias, ls, resid, nout


def _run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi):
    import ctypes
    from aigv_assessor_amd.native import ptr
    M, K, N = A.shape[0], A.shape[1], W.shape[0]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    dA, dW = dev(A), dev(W)
    dC = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
    db, dl, dr = (dev(t) if t is not None else None for t in (bias, ls, resid))
    sync(lib.aigv_op_gemm_rows(ptr(dA), K, ptr(dW), K, ptr(dC), nout, ptr(db), ptr(dl), ptr(dr), nout, (ctypes.c_int32 * len(cu))(*cu), len(lens),
                               N, K, epi, None), lib)
    return dC.cpu()


# sequences: whole tiles only; the benched clip (2176 = 8 x 256 + 128); ragged tails of one and two halves; InternViT frames (1025 =
# 4 x 256 + 1: tiny tails on the skinny kernel, uniform and not); sequences shorter than a half tile; N = 256 j + 128 (column split)
@pytest.mark.parametrize("lens,N,K,epi", per_epilogue([([512, 256], 256, 128), ([2176, 2176, 2176], 512, 1024), ([300, 77, 1000, 129, 511], 768, 512),
                                                      ([1025] * 5, 512, 1024), ([1025, 1027, 258, 3], 256, 256), ([40, 5, 130], 384, 192),
                                                      ([2176] * 4, 1024, 3584)]))
def test_gemm_row_plan_epilogues(lib, lens, N, K, epi):
    """aigv_op_gemm_rows - the scoring pass's dispatch: per-sequence body tiles through the half-tile table, (ragged) tail halves as
    K slices with a factor fixed by (N, K), tiny tails on the skinny kernel - against the rounded fp32 reference."""
    g = torch.Generator().manual_seed(sum(lens) * 7 + N + K + epi)
    A, W, bias, ls, resid, nout = _gemm_rows_case(g, lens, N, K, epi)
    want = gemm_ref(A, W, epi, bias, ls, resid)
    got = _run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
    ulp_check(got, want, frac=0.03 if epi in (1, 4) else 0.02, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)


@pytest.mark.parametrize("N,K,epi", per_epilogue([(512, 1024), (1024, 3584), (384, 256)]))
def test_gemm_row_plan_is_batch_invariant(lib, N, K, epi):
    """A sequence's rows give the same BITS whatever the other sequences of the call are: every sequence of a mixed
    batch alone == inside the batch, for body rows, split-K tail halves (one and two, ragged) and tiny tails."""
    lens = [2176, 1025, 300, 2176, 131, 1025, 2]
    g = torch.Generator().manual_seed(N + K + epi)
    A, W, bias, ls, resid, nout = _gemm_rows_case(g, lens, N, K, epi)
    both = _run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
    assert torch.isfinite(both.float()).all()
    r0 = 0
    for n in lens:
        sl = slice(r0, r0 + n)
        one = _run_gemm_rows(lib, A[sl], W, bias, ls, None if resid is None else resid[sl], nout, [n], epi)
        assert torch.equal(one.view(torch.int16), both[sl].view(torch.int16)), f"sequence of {n} rows at {r0}"
        r0 += n
    # and as a uniform batch of the same sequence (the tiny tails then share one strided skinny launch)
    for n in (1025, 2176):
        start = sum(lens[:lens.index(n)])
        sl = slice(start, start + n)
        rep = 3
        A3 = A[sl].repeat(rep, 1)
        r3 = None if resid is None else resid[sl].repeat(rep, 1)
        got = _run_gemm_rows(lib, A3, W, bias, ls, r3, nout, [n] * rep, epi)
        for i in range(rep):
            assert torch.equal(got[i * n:(i + 1) * n].view(torch.int16), both[sl].view(torch.int16))


@pytest.mark.parametrize("lens,N,K,epi", per_epilogue([([2176], 4096, 4096), ([2176], 6144, 4096), ([2176, 2176], 4096, 14336), ([300, 2176, 641], 1024, 3584), ([2233], 512, 2048)]))
def test_gemm_row_plan_fused_tail_slices_change_no_bit(lib, lens, N, K, epi):
    """One or two clips leave the body's last round of CUs part empty; the tail tiles' K slices then run INSIDE the body's launch
    (gemm256_kernel<.., FUSE>; AIGV_TUNE_FUSE_TAILS 0 = by fill, 1 = never, 2 = always).  Same slices, same slabs, same finalize pass:
    the three settings give the same bits - which is why the choice may follow the batch's fill without touching batch invariance."""
    from aigv_assessor_amd import native
    g = torch.Generator().manual_seed(sum(lens) + N + K + epi)
    A, W, bias, ls, resid, nout = _gemm_rows_case(g, lens, N, K, epi)
    outs = []
    try:
        for fuse, lone in ((1, 1), (2, 1), (0, 1), (1, 2), (0, 0)):   # AIGV_TUNE_FUSE_TAILS (12) x AIGV_TUNE_LONE_BODY (13: the body on gemmco.hip's LONE form)
            native.check(lib.aigv_tune_default(12, fuse))
            native.check(lib.aigv_tune_default(13, lone))
            outs.append(_run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi))
    finally:
        native.check(lib.aigv_tune_default(12, 0))
        native.check(lib.aigv_tune_default(13, 1))      # (the library's default: never)
    assert torch.isfinite(outs[0].float()).all()
    for o in outs[1:]:
        assert torch.equal(outs[0].view(torch.int16), o.view(torch.int16))
    ulp_check(outs[1], gemm_ref(A, W, epi, bias, ls, resid), frac=0.03, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)


@pytest.mark.parametrize("lens,N,K,epi", per_epilogue([([2176], 512, 1024), ([1025] * 8, 1024, 1024), ([2176, 300, 1025], 1024, 3584), ([2233], 4096, 4096)]))
def test_gemm_row_plan_body_tile_changes_no_bit(lib, lens, N, K, epi):
    """A row plan's body rows can run on the 256x256 kernel (shipped) or on the 128x128 kernel through the same half-tile table.  Both sum
    every output element over the full K in the same order, so the two give the same BITS - the fact behind 'gemm modes 1 and 2 are
    batch-invariant and identical to the default on body rows': forced 256 == forced 128 == default."""
    from aigv_assessor_amd import native
    g = torch.Generator().manual_seed(sum(lens) + N + K + epi)
    A, W, bias, ls, resid, nout = _gemm_rows_case(g, lens, N, K, epi)
    outs = []
    try:
        for body_tile in (1, 2, 0):
            native.check(lib.aigv_tune_gemm(0 + 32 + (body_tile << 14), 0.0))
            outs.append(_run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi))
    finally:
        native.check(lib.aigv_tune_gemm(0 + 32, 0.0))
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)) and torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))


@pytest.mark.parametrize("M,N,K,epi,mode", per_epilogue([(1100, 512, 448, 0), (777, 256, 64, 1), (515, 768, 1024, 2), (300, 256, 192, 3),
                                                        (1029, 1024, 512, 4), (256, 256, 128, 0),
                                                        (12000, 1408, 256, 0), (12100, 1408, 320, 2), (600, 384, 256, 3)],   # N = 256 j + 128: column split on the big ones
                                                       second=(1, 2 + 16, 2 + 32, 2 + 64, 0)))
def test_gemm_tile_kernels_agree(lib, mode, M, N, K, epi):
    """mode 1 = 128x128 kernel, 2 + 16*(1+v) = 256x256 phase-interleaved kernel with schedule variant v,
    0 = cost-model split (256 main + 128/skinny tail)."""
    from aigv_assessor_amd import native
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) * (1.0 / math.sqrt(K))).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g) + 0.5).to(BF) if epi == 2 else None
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g).to(BF) if epi in (2, 3) else None
    want = gemm_ref(A, W, epi, bias, ls, resid)
    dA, dW = dev(A), dev(W)
    dC = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
    db, dl, dr = (dev(t) if t is not None else None for t in (bias, ls, resid))
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    try:
        sync(lib.aigv_op_gemm(ptr(dA), K, ptr(dW), K, ptr(dC), nout, ptr(db), ptr(dl), ptr(dr), nout, None, 0, M, N, K, epi,
                              None), lib)
    finally:
        native.check(lib.aigv_tune_gemm(0 + 32, 0.0))      # back to auto + the default schedule (variant 1)
    ulp_check(dC, want, frac=0.03, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)


def _op_gemm_mode(lib, mode, co_kmax, A, W, epi, bias, ls, resid, pos=None, np_=0, rows_out=None):
    """aigv_op_gemm under process GEMM mode ``mode`` and co-resident threshold ``co_kmax``; defaults restored afterwards."""
    from aigv_assessor_amd import native
    from aigv_assessor_amd.native import ptr
    M, K, N = A.shape[0], A.shape[1], W.shape[0]
    nout = N // 2 if epi == 4 else N
    dC = torch.full((rows_out or M, nout), float("nan"), dtype=BF, device="cuda")
    db, dl, dr, dp = (dev(t) if t is not None else None for t in (bias, ls, resid, pos))
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    native.check(lib.aigv_tune_co_gemm(co_kmax))
    try:
        sync(lib.aigv_op_gemm(ptr(dev(A)), K, ptr(dev(W)), K, ptr(dC), nout, ptr(db), ptr(dl), ptr(dr), nout, ptr(dp), np_, M, N, K, epi, None), lib)
    finally:
        native.check(lib.aigv_tune_gemm(0 + 32, 0.0))
        native.check(lib.aigv_tune_co_gemm(0))
    return dC.cpu()


# InternViT's four linears at a few frames (qkv 3072, proj 1024, fc1 4096 out of K = 1024; fc2 K = 4096), ragged row counts (a last
# tile with one valid half, with 1 row, with 129 rows), K-tile counts 1, 2, 3 (prologue / tail paths of the ring), odd and even, N = 128
@pytest.mark.parametrize("M,N,K,epi", per_epilogue([(2050, 3072, 1024), (1025, 1024, 1024), (1300, 4096, 1024), (1153, 1024, 4096), (256, 128, 64), (257, 256, 128),
                                                   (129, 384, 192), (4099, 512, 448), (77, 640, 1088)]))
def test_gemm_co_resident_kernel_equals_the_256_kernel_bit_for_bit(lib, M, N, K, epi):
    """gemmco.hip (256 x 128 tile, two workgroups per CU) against gemm256.hip / gemm.hip: the same MFMA chain per output element and the
    same epilogue rounding points, so torch.equal - the property that lets the dispatcher move InternViT's K = 1024 linears onto it
    without moving one recorded bit (reference semantics: modeling_intern_vit.py:192-228)."""
    g = torch.Generator().manual_seed(M * 3 + N + K + epi)
    A, W, bias, ls, resid, nout = _gemm_rows_case(g, [M], N, K, epi)
    co = _op_gemm_mode(lib, 4, 0, A, W, epi, bias, ls, resid)
    assert torch.isfinite(co.float()).all()
    ref = _op_gemm_mode(lib, 2 if N % 256 == 0 else 1, 0, A, W, epi, bias, ls, resid)
    assert torch.equal(co.view(torch.int16), ref.view(torch.int16)), f"{int((co.view(torch.int16) != ref.view(torch.int16)).sum())} elements differ"
    # its other two schedules: reads / requests / MFMAs in blocks (the first version), and the LONE form for launches of at most one workgroup
    # per CU (eight waves, four of them only issue the LDS-DMA requests) - mode word 4 + 16 * (1 + variant)
    for word in (4 + 16, 4 + 16 * 7):
        other = _op_gemm_mode(lib, word, 0, A, W, epi, bias, ls, resid)
        assert torch.equal(other.view(torch.int16), ref.view(torch.int16)), f"schedule word {word}: {int((other.view(torch.int16) != ref.view(torch.int16)).sum())} elements differ"
    ulp_check(co, gemm_ref(A, W, epi, bias, ls, resid), frac=0.03, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)
    # and the default dispatch (mode 0) takes it for K <= 1024 and only then: same bits either way
    auto = _op_gemm_mode(lib, 0, 1024, A, W, epi, bias, ls, resid)
    if K <= 1024:
        assert torch.equal(auto.view(torch.int16), co.view(torch.int16))


def test_gemm_co_resident_kernel_patch_epilogue(lib):
    """The patch-embedding epilogue (bias, position rows, one skipped class row per frame; K = 640 as at 448 px / patch 14) on the
    co-resident kernel == on the 256 kernel."""
    g = torch.Generator().manual_seed(11)
    F_, np_, N, K = 3, 1024, 1024, 640
    M = F_ * np_
    A, W = (torch.randn(M, K, generator=g) * 0.5).to(BF), (torch.randn(N, K, generator=g) * 0.1).to(BF)
    bias, pos = (torch.randn(N, generator=g) * 0.1).to(BF), torch.randn(np_ + 1, N, generator=g).to(BF)
    outs = [_op_gemm_mode(lib, mode, 0, A, W, 5, bias, None, None, pos, np_, rows_out=F_ * (np_ + 1)).view(F_, np_ + 1, N)[:, 1:] for mode in (4, 2)]
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].reshape(M, N).view(torch.int16), outs[1].reshape(M, N).view(torch.int16))
    ulp_check(outs[0].reshape(M, N), gemm_ref(A, W, 5, bias=bias, pos=pos, np_=np_), atol_rel=2.0 ** -7)


@pytest.mark.parametrize("lens,N,K,epi", per_epilogue([([1025] * 8, 3072, 1024), ([1025, 1027, 258, 3], 256, 256), ([300, 77, 1000, 129, 511], 768, 512), ([2176, 131], 1024, 1024)]))
def test_gemm_row_plan_on_the_co_resident_kernel(lib, lens, N, K, epi):
    """aigv_op_gemm_rows with K <= the co-resident threshold (aigv_tune_co_gemm; off by default): body AND tail half tiles of every sequence in ONE launch of gemmco.hip
    through the half-tile table (ragged halves, halves of different sequences sharing a tile), tiny tails on the skinny kernel.  Every
    sequence alone == inside the batch (bits), and the body / tail rows equal the full-K tile kernels' bits."""
    from aigv_assessor_amd import native
    g = torch.Generator().manual_seed(sum(lens) + N + K + epi)
    A, W, bias, ls, resid, nout = _gemm_rows_case(g, lens, N, K, epi)
    native.check(lib.aigv_tune_co_gemm(1024))                                     # (the shipped default is 0 = never: profiles/r5_gemmco.txt)
    try:
        both = _run_gemm_rows(lib, A, W, bias, ls, resid, nout, lens, epi)
        ulp_check(both, gemm_ref(A, W, epi, bias, ls, resid), frac=0.03, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)
        r0 = 0
        for n in lens:
            sl = slice(r0, r0 + n)
            one = _run_gemm_rows(lib, A[sl], W, bias, ls, None if resid is None else resid[sl], nout, [n], epi)
            assert torch.equal(one.view(torch.int16), both[sl].view(torch.int16)), f"sequence of {n} rows at {r0}"
            r0 += n
    finally:
        native.check(lib.aigv_tune_co_gemm(0))
    full = _op_gemm_mode(lib, 2 if N % 256 == 0 else 1, 0, A, W, epi, bias, ls, resid)   # every row on one tile kernel, full K
    r0 = 0
    for n in lens:
        keep = n - (n % 256 if n % 256 <= 4 else 0)                                # tiny tails run on the skinny kernel (another summation order)
        assert torch.equal(full[r0:r0 + keep].view(torch.int16), both[r0:r0 + keep].view(torch.int16)), f"sequence of {n} rows at {r0}"
        r0 += n


@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm_tile_order_changes_no_bit(lib, epi):
    """The 256x256 kernel's workgroup -> tile map (row groups / groups of g column tiles: GemmArgs::order, gemm256.hip) only decides WHICH
    workgroup computes a tile, never how: every order must produce the same bits.  8708 x 1280 x 512 = 35 row tiles (the last with 4 rows)
    x 5 column tiles on the 256 kernel for every row (mode 2): with groups of 4 and of 2 column tiles the last column group is ragged
    (5 % 4, 5 % 2 != 0), with row groups of 4 the last row group is (35 % 4 != 0).  The default rule switches to column groups only for
    N x K >= 32 Mi with M >= 8192 - InternLM2's w1|w3 / w2 at batch 4 - which no other op-level case reaches."""
    from aigv_assessor_amd import native
    from aigv_assessor_amd.native import ptr
    M, N, K = 8708, 1280, 512
    g = torch.Generator().manual_seed(4100 + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) * (1.0 / math.sqrt(K))).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g) + 0.5).to(BF) if epi == 2 else None
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g).to(BF) if epi in (2, 3) else None
    dA, dW = dev(A), dev(W)
    db, dl, dr = (dev(t) if t is not None else None for t in (bias, ls, resid))
    outs = {}
    try:
        for field in (1, 5, 3, 2):     # aigv_tune_gemm bits 10..13: 1 = row groups, 1 + g = groups of g column tiles (4, 2, 1)
            native.check(lib.aigv_tune_gemm(2 | (field << 10), 0.0))
            dC = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
            sync(lib.aigv_op_gemm(ptr(dA), K, ptr(dW), K, ptr(dC), nout, ptr(db), ptr(dl), ptr(dr), nout, None, 0, M, N, K, epi, None), lib)
            outs[field] = dC.cpu()
    finally:
        native.check(lib.aigv_tune_gemm(0, 0.0))           # auto dispatch, tile order by weight size again
    ulp_check(outs[1], gemm_ref(A, W, epi, bias, ls, resid), frac=0.03, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)
    for field in (5, 3, 2):
        assert torch.equal(outs[field].view(torch.int16), outs[1].view(torch.int16)), f"tile order field {field} changed the result"


@pytest.mark.parametrize("M,N,K,epi,S", [(516, 512, 2048, 3, 2), (200, 256, 3072, 0, 3), (4, 1024, 4096, 4, 4), (300, 384, 1024, 2, 2),
                                          (130, 256, 512, 1, 4), (512, 512, 2048, 3, -8), (768, 256, 1536, 0, -3), (300, 1024, 1024, 4, -2),
                                          (256, 768, 4096, 2, -4), (513, 256, 768, 1, -6)])
def test_gemm_splitk_tail(lib, M, N, K, epi, S):
    """Split-K tail path: fp32 slabs + fixed-order finalize with the same epilogues.  S < 0: |S| slices from the 256x256
    kernel (whole row tiles that would not fill a round), S > 0: from the 128x128 kernel."""
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(M + N + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) * (1.0 / math.sqrt(K))).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (0, 1, 2) else None
    ls = (torch.rand(N, generator=g) + 0.5).to(BF) if epi == 2 else None
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g).to(BF) if epi in (2, 3) else None
    want = gemm_ref(A, W, epi, bias, ls, resid)
    dC = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
    ws = torch.empty(abs(S) * M * N, dtype=torch.float32, device="cuda")
    db, dl, dr = (dev(t) if t is not None else None for t in (bias, ls, resid))
    op = lib.aigv_op_gemm_splitk if S > 0 else lib.aigv_op_gemm_splitk256
    sync(op(ptr(dev(A)), K, ptr(dev(W)), K, ptr(dC), nout, ptr(db), ptr(dl), ptr(dr), nout, M, N, K, epi, abs(S), ptr(ws), None), lib)
    ulp_check(dC, want, frac=0.03, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)


@pytest.mark.parametrize("mode", [1, 2])
def test_gelu_epilogue_over_all_bf16_inputs(lib, mode):
    """The epilogue's GELU (one polynomial + one exp2 instead of libm erff) against torch's CPU bf16 GELU - the
    reference's op (modeling_intern_vit.py:243 ACT2FN['gelu'], modeling_internvl_chat.py:85 nn.GELU) - on EVERY bf16
    input: the GELU argument is a bf16 value, so this is the whole domain.  A one-hot W makes C = gelu(A) exactly.
    Tolerance: identical bf16 results except <= 12 one-ulp differences for 2^-60 <= |x| <= 4 and x > 4; for
    x < -4 (|gelu| < 1.3e-4, where the reference's own 1 + erf has cancelled to a few bits) 2^-17 absolute."""
    from aigv_assessor_amd import native
    from aigv_assessor_amd.native import ptr
    bits = torch.arange(65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    x = torch.where(torch.isfinite(x), x, torch.zeros_like(x)).to(BF)
    A = x.reshape(1024, 64)
    W = torch.zeros(256, 64)
    W[torch.arange(64), torch.arange(64)] = 1.0
    dC = torch.empty(1024, 256, dtype=BF, device="cuda")
    native.check(lib.aigv_tune_gemm(mode, 0.0))
    try:
        sync(lib.aigv_op_gemm(ptr(dev(A)), 64, ptr(dev(W.to(BF))), 64, ptr(dC), 256, None, None, None, 0, None, 0, 1024, 256, 64, 1, None), lib)
    finally:
        native.check(lib.aigv_tune_gemm(0, 0.0))
    got = dC.cpu()[:, :64].reshape(-1).float()
    assert torch.equal(dC.cpu()[:, 64:].float(), torch.zeros(1024, 192))
    want = torch.nn.functional.gelu(x).float()
    xf = x.float()
    main = ((xf.abs() >= 2.0 ** -60) & (xf >= -4.0) & (xf.abs() < 1e38)) | (xf == 0)
    diff = (got - want).abs()
    ulp = want.abs().clamp_min(1e-38).log2().floor().exp2() * 2.0 ** -7
    assert int((diff[main] != 0).sum()) <= 12, int((diff[main] != 0).sum())
    assert bool((diff[main] <= ulp[main]).all())
    tail = (xf < -4.0) & torch.isfinite(xf)
    assert float(diff[tail].max()) <= 2.0 ** -17
    tiny = (xf.abs() < 2.0 ** -60) & (xf != 0)          # results far below bf16's useful range: 0.5 x either way or flushed
    assert float(diff[tiny].max()) <= 2.0 ** -60


def test_gemm_identity_asymmetric(lib):
    """A = I with an asymmetric W catches a transposed C write (guide §3)."""
    from aigv_assessor_amd.native import ptr
    n = 128
    A = torch.eye(n).to(BF)
    W = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(BF)   # W[i][j] != W[j][i]
    dC = torch.empty(n, n, dtype=BF, device="cuda")
    sync(lib.aigv_op_gemm(ptr(dev(A)), n, ptr(dev(W)), n, ptr(dC), n, None, None, None, 0, None, 0, n, n, n, 0, None), lib)
    assert torch.equal(dC.cpu().float(), W.float().t())


def test_gemm_inplace_residual_and_patch_epilogue(lib):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(3)
    # in-place residual (C aliases resid), as the ViT/LLM layers use it
    M, N, K = 514, 256, 128
    A, W = (torch.randn(M, K, generator=g) * 0.5).to(BF), (torch.randn(N, K, generator=g) * 0.1).to(BF)
    x = torch.randn(M, N, generator=g).to(BF)
    want = gemm_ref(A, W, 3, resid=x)
    dx = dev(x.clone())
    sync(lib.aigv_op_gemm(ptr(dev(A)), K, ptr(dev(W)), K, ptr(dx), N, None, None, ptr(dx), N, None, 0, M, N, K, 3, None), lib)
    ulp_check(dx, want, atol_rel=2.0 ** -7)
    # patch epilogue: bias, position rows, one skipped class row per frame
    F_, np_, N, K = 3, 128, 128, 64
    M = F_ * np_
    A, W = (torch.randn(M, K, generator=g) * 0.5).to(BF), (torch.randn(N, K, generator=g) * 0.1).to(BF)
    bias, pos = (torch.randn(N, generator=g) * 0.1).to(BF), torch.randn(np_ + 1, N, generator=g).to(BF)
    want = gemm_ref(A, W, 5, bias=bias, pos=pos, np_=np_)
    dC = torch.zeros(F_ * (np_ + 1), N, dtype=BF, device="cuda")
    sync(lib.aigv_op_gemm(ptr(dev(A)), K, ptr(dev(W)), K, ptr(dC), N, ptr(dev(bias)), None, None, 0, ptr(dev(pos)), np_, M, N,
                          K, 5, None), lib)
    got = dC.cpu().view(F_, np_ + 1, N)
    assert (got[:, 0] == 0).all()            # class rows untouched
    ulp_check(got[:, 1:].reshape(M, N), want, atol_rel=2.0 ** -7)


def test_gemm_rejects_bad_shapes(lib):
    from aigv_assessor_amd import native
    a = torch.zeros(128, 64, dtype=BF, device="cuda")
    rc = lib.aigv_op_gemm(native.ptr(a), 64, native.ptr(a), 64, native.ptr(a), 100, None, None, None, 0, None, 0, 128, 100, 64, 0, None)
    assert rc == -1 and b"multiple of 128" in lib.aigv_last_error(None)
    rc = lib.aigv_op_gemm(native.ptr(a), 64, native.ptr(a), 64, native.ptr(a), 128, None, None, None, 0, None, 0, 128, 128, 40, 0, None)
    assert rc == -1


# ---------------------------------------------------------------------------------------------------------
# row ops
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,H", [(37, 1024), (5, 4096), (3, 2304), (9, 128)])
def test_layernorm(lib, rows, H):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(rows + H)
    x = (torch.randn(rows, H, generator=g) * 2 + 0.3).to(BF)
    w, b = (1 + 0.1 * torch.randn(H, generator=g)).to(BF), (0.1 * torch.randn(H, generator=g)).to(BF)
    want = torch.nn.functional.layer_norm(x.float(), (H,), w.float(), b.float(), 1e-6).to(BF)
    y = torch.empty_like(x, device="cuda")
    sync(lib.aigv_op_layernorm(ptr(dev(x)), H, ptr(dev(w)), ptr(dev(b)), ptr(y), H, rows, H, 1e-6, None), lib)
    ulp_check(y, want.float())


@pytest.mark.parametrize("rows,H", [(33, 4096), (7, 512)])
def test_rmsnorm_and_row_gather(lib, rows, H):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, H, generator=g) * 3).to(BF)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(BF)

    def ref(xx):
        xf = xx.float()
        return rb(w.float() * rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)))
    y = torch.empty_like(x, device="cuda")
    sync(lib.aigv_op_rmsnorm(ptr(dev(x)), H, ptr(dev(w)), ptr(y), H, rows, H, 1e-5, None, None), lib)
    ulp_check(y, ref(x))
    idx = torch.tensor([rows - 1, 0, 2], dtype=torch.int32)
    y2 = torch.empty(3, H, dtype=BF, device="cuda")
    sync(lib.aigv_op_rmsnorm(ptr(dev(x)), H, ptr(dev(w)), ptr(y2), H, 3, H, 1e-5, ptr(dev(idx)), None), lib)
    ulp_check(y2, ref(x[idx.long()]))


def test_rope_matches_reference_rounding(lib):
    from aigv_assessor_amd.native import ptr
    from aigv_assessor_amd.modeling import rope_tables
    g = torch.Generator().manual_seed(9)
    T, nkv, grp, D = 50, 2, 2, 128            # 2 kv groups x (2 q + K + V)
    ld = nkv * (grp + 2) * D
    qkv = torch.randn(T, ld, generator=g).to(BF)
    pos = torch.randint(0, 300, (T,), generator=g, dtype=torch.int32)
    cos, sin = rope_tables(D, 1e6, 300)
    want = qkv.clone().view(T, nkv, grp + 2, D)
    c = torch.cat([cos, cos], -1)[pos.long()][:, None, None, :]
    s = torch.cat([sin, sin], -1)[pos.long()][:, None, None, :]
    x = want[:, :, : grp + 1, :]
    rot = torch.cat((-x[..., D // 2:], x[..., : D // 2]), dim=-1)
    want[:, :, : grp + 1, :] = (x * c) + (rot * s)       # bf16 ops = the reference's three rounding points
    d = dev(qkv.clone())
    sync(lib.aigv_op_rope(ptr(d), ld, ptr(dev(pos)), ptr(dev(cos)), ptr(dev(sin)), T, grp + 1, grp + 2, nkv, D, None), lib)
    assert torch.equal(d.cpu().view(T, nkv, grp + 2, D), want)


def test_pixel_shuffle_and_im2col_are_bit_exact(lib):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(4)
    F_, grid, Hv = 3, 6, 64
    vit = torch.randn(F_, grid * grid + 1, Hv, generator=g).to(BF)
    x = vit[:, 1:].reshape(F_, grid, grid, Hv)
    want = x.reshape(F_, grid // 2, 2, grid // 2, 2, Hv).permute(0, 1, 3, 2, 4, 5).reshape(F_, (grid // 2) ** 2, 4 * Hv)
    out = torch.empty(F_, (grid // 2) ** 2, 4 * Hv, dtype=BF, device="cuda")
    sync(lib.aigv_op_pixel_shuffle(ptr(dev(vit)), grid, Hv, ptr(out), F_, None), lib)
    assert torch.equal(out.cpu(), want)
    S, P, Kp = 56, 14, 640
    fr = torch.randn(2, 3, S, S, generator=g).to(BF)
    want = torch.nn.functional.unfold(fr.float(), P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P).to(BF)
    col = torch.full((2 * 16, Kp), 7.0, dtype=BF, device="cuda")
    sync(lib.aigv_op_im2col(ptr(dev(fr)), 2, 3, S, P, Kp, ptr(col), None), lib)
    assert torch.equal(col.cpu()[:, : 3 * P * P], want) and (col.cpu()[:, 3 * P * P:] == 0).all()


# ---------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------
def attn_truth(q, k, v, causal, scale_pre, post_div, dtype):
    """q [n,h,d], k/v [n,hk,d] for ONE sequence.  dtype=float64 -> truth; bf16 -> the reference's eager path
    (modeling_intern_vit.py:153-157 / modeling_internlm2.py:407-424) with its rounding points."""
    h, hk = q.shape[1], k.shape[1]
    rep = h // hk
    qq = q.transpose(0, 1).to(dtype)
    kk = k.transpose(0, 1).repeat_interleave(rep, 0).to(dtype)
    vv = v.transpose(0, 1).repeat_interleave(rep, 0).to(dtype)
    if scale_pre != 1.0:
        qq = qq * scale_pre
    s = qq @ kk.transpose(1, 2)
    if post_div != 1.0:
        s = s / post_div
    if causal:
        n = q.shape[0]
        m = torch.full((n, n), torch.finfo(dtype).min, dtype=dtype).triu(1)
        s = s + m
    if dtype == BF and post_div != 1.0:
        p = torch.softmax(s, -1, dtype=torch.float32).to(BF)      # LLM: fp32 softmax, cast back
    else:
        p = torch.softmax(s, -1)
    return (p @ vv).transpose(0, 1)


def run_attention(lib, q, k, v, lens, causal, d, pre, post, uniform=False, round_scores=False, lead_key=False):
    from aigv_assessor_amd.native import ptr
    T, h, hk = q.shape[0], q.shape[1], k.shape[1]
    g = h // hk
    # fused layout like the LLM's wqkv output: per kv group [g q heads | K | V]
    fused = torch.zeros(T, hk, g + 2, d, dtype=BF)
    fused[:, :, :g] = q.view(T, hk, g, d)
    fused[:, :, g] = k
    fused[:, :, g + 1] = v
    dq = dev(fused.view(T, -1))
    ld = hk * (g + 2) * d
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    out = torch.full((T, h * d), float("nan"), dtype=BF, device="cuda")
    base = dq.data_ptr()
    sync(lib.aigv_op_attention(base, ld, base + g * d * 2, ld, base + (g + 1) * d * 2, ld, ptr(out), h * d, ptr(dev(cu)),
                               len(lens), max(lens), h, hk, (g + 2) * d, (g + 2) * d, d, int(causal) | (2 if uniform else 0) | (4 if round_scores else 0) | (8 if lead_key else 0),
                               post, pre, None), lib)
    return out.cpu().view(T, h, d)


# kernel (aigv_tune_attention): 0 = the default form (4 waves per workgroup, two-deep K/V ring; with the uniform-length hint where the
# lengths are equal, so that the 1025-row ViT shape runs its left-over row in the key-split form), 8 = 8 waves per workgroup
# (the form kept for A/B)
@pytest.mark.parametrize("kernel", [0, 8])
@pytest.mark.parametrize("d,causal,h,hk,lens", [
    (64, False, 2, 2, [1025, 1025, 1025]),       # ViT: 448 px frames, cls tail row
    (64, False, 3, 3, [257, 257]),               # ViT: 224 px
    (64, False, 2, 2, [512, 300, 33]),           # ragged non-causal: full, partial and one-sub-block query blocks
    (128, True, 4, 2, [200, 77]),                # LLM: GQA, ragged clips
    (128, True, 8, 2, [513, 64, 1]),             # group of 4, tile-boundary lengths, length-1 clip
    (128, True, 2, 1, [1300]),
    (128, True, 8, 2, [2176]),                   # the canonical clip: 17 x 128 rows, 34 key tiles
    (128, False, 2, 2, [384, 129]),              # InternViT-6B head width, non-causal
])
@pytest.mark.parametrize("round_scores", [True, False])
def test_attention_matches_eager_reference(lib, d, causal, h, hk, lens, kernel, round_scores):
    """round_scores: the score matrix rounded to bf16 where the reference's eager path rounds it (the default of the scoring pass since
    round 4) - the kernel must then sit much closer to the EAGER bf16 result than that result sits to fp64 truth (the rounding noise of
    the scores is shared); False: fp32 scores (rounds 1-3), at least as accurate against fp64 truth as the eager path."""
    sync(lib.aigv_tune_attention(kernel), lib)
    try:
        _attention_case(lib, d, causal, h, hk, lens, uniform=(kernel == 0 and len(set(lens)) == 1), round_scores=round_scores)
    finally:
        sync(lib.aigv_tune_attention(0), lib)


def _attention_case(lib, d, causal, h, hk, lens, uniform, round_scores=False):
    g = torch.Generator().manual_seed(sum(lens) + d)
    T = sum(lens)
    q = (torch.randn(T, h, d, generator=g) * 1.5).to(BF)
    k = (torch.randn(T, hk, d, generator=g) * 1.5).to(BF)
    v = torch.randn(T, hk, d, generator=g).to(BF)
    # force large, late-arriving maxima so the online-softmax rescale path is exercised (guide rule 26)
    k[lens[0] // 2] *= 6.0
    pre = d ** -0.5 if not causal else 1.0
    post = 1.0 if not causal else math.sqrt(d)
    got = run_attention(lib, q, k, v, lens, causal, d, pre, post, uniform, round_scores).double()
    off = 0
    near, far = 0.0, 0.0
    for n in lens:
        sl = slice(off, off + n)
        truth = attn_truth(q[sl], k[sl], v[sl], causal, pre, post, torch.float64)
        eager = attn_truth(q[sl], k[sl], v[sl], causal, pre, post, BF).double()
        e_hip = (got[sl] - truth).abs()
        e_ref = (eager - truth).abs()
        assert torch.isfinite(got[sl]).all()
        assert e_hip.mean() <= 1.5 * e_ref.mean() + 1e-4, (e_hip.mean().item(), e_ref.mean().item())
        assert e_hip.max() <= 2.0 * e_ref.max() + 2e-3, (e_hip.max().item(), e_ref.max().item())
        near += (got[sl] - eager).abs().sum().item()
        far += e_ref.sum().item()
        off += n
    print(f"sum |hip - eager bf16| / sum |eager bf16 - fp64 truth| = {near / max(far, 1e-30):.3f} (round_scores={round_scores})")
    if round_scores:       # what remains is the P rounding (un-normalised here, normalised there) and the output rounding
        assert near <= 0.75 * far, (near, far)


@pytest.mark.parametrize("d,h,lens", [(64, 2, [1025, 1025]), (64, 3, [257, 65, 1]), (128, 2, [1025, 129])])
@pytest.mark.parametrize("round_scores", [True, False])
def test_attention_lead_key_form_equals_the_plain_loop_up_to_rounding(lib, d, h, lens, round_scores):
    """Non-causal key counts 64 j + 1 (InternViT: cls + 1024 patches), the opt-in lead-key form: the loop runs over 16 full, unmasked tiles
    and key 0 is merged in the epilogue, instead of 16 tiles + one tile with a single live key.  Same softmax, another fp32 summation
    order: within one bf16 ulp of the row scale of the default form (incl. the key-split block of the 1025th row and a one-key sequence)."""
    g = torch.Generator().manual_seed(sum(lens) + d + h)
    T = sum(lens)
    q = (torch.randn(T, h, d, generator=g) * 1.5).to(BF)
    k = (torch.randn(T, h, d, generator=g) * 1.5).to(BF)
    v = torch.randn(T, h, d, generator=g).to(BF)
    k[0] *= 4.0           # a dominant lead key in the first sequence: its weight decides most rows
    a = run_attention(lib, q, k, v, lens, False, d, d ** -0.5, 1.0, False, round_scores, lead_key=True).float()
    b = run_attention(lib, q, k, v, lens, False, d, d ** -0.5, 1.0, False, round_scores, lead_key=False).float()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    # outputs are weighted means of O(1) values: an element's rounding noise is a bf16 ulp of the ROW's scale, not of its own (possibly tiny) value
    scale = b.abs().amax(-1, keepdim=True).clamp_min(1e-3)
    assert ((a - b).abs() <= 2.0 ** -7 * scale).all(), ((a - b).abs() / scale).max().item()
    assert (a - b).abs().mean().item() <= 2.0 ** -10 * b.abs().mean().item()
    assert (a != b).float().mean().item() <= 0.25


def test_attention_with_fused_query_rope_equals_rope_then_attention(lib):
    """Query RoPE inside the attention kernel's Q load (K rotated in memory by the rope kernel on its slot only) is
    bit-identical to rotating q and k in memory first: the same three bf16 roundings, only the place differs."""
    from aigv_assessor_amd.native import ptr
    d, h, hk, lens = 128, 8, 2, [300, 77, 129]
    g = h // hk
    T, ld = sum(lens), hk * (g + 2) * d
    gen = torch.Generator().manual_seed(3)
    qkv = torch.randn(T, ld, generator=gen).to(BF)
    pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32)
    ang = torch.arange(0, 400)[:, None].float() * (1.0 / (10000 ** (torch.arange(0, d // 2).float() / (d // 2))))[None, :]
    cos, sin = ang.cos().to(BF), ang.sin().to(BF)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    dcu, dpos, dcos, dsin = dev(cu), dev(pos), dev(cos), dev(sin)
    outs = []
    for fused in (False, True):
        x = dev(qkv.clone())
        base = x.data_ptr()
        # slots per group: g query heads, K, V
        sync(lib.aigv_op_rope(base, ld, ptr(dpos), ptr(dcos), ptr(dsin), T, 1 if fused else g + 1, g + 2, hk, d, None) if not fused
             else lib.aigv_op_rope(base + g * d * 2, ld, ptr(dpos), ptr(dcos), ptr(dsin), T, 1, g + 2, hk, d, None), lib)
        out = torch.full((T, h * d), float("nan"), dtype=BF, device="cuda")
        args = (base, ld, base + g * d * 2, ld, base + (g + 1) * d * 2, ld, ptr(out), h * d, ptr(dcu), len(lens), max(lens), h, hk,
                (g + 2) * d, (g + 2) * d, d, 1, math.sqrt(d), 1.0)
        if fused:
            sync(lib.aigv_op_attention_rope(*args, ptr(dpos), ptr(dcos), ptr(dsin), None), lib)
        else:
            sync(lib.aigv_op_attention(*args, None), lib)
        outs.append(out.cpu())
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


# ---------------------------------------------------------------------------------------------------------
# skinny GEMM / lm-head argmax
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,N,K,epi", [(5, 272, 256, 0), (16, 512, 384, 1), (33, 256, 1152, 3), (64, 640, 256, 2), (1, 128, 128, 0)])
def test_skinny_gemm(lib, R, N, K, epi):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(R + N + K)
    x = (torch.randn(R, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (0, 3) else None
    nout = N // 2 if epi == 2 else N
    resid = torch.randn(R, nout, generator=g).to(BF) if epi == 1 else None
    want = gemm_ref(x, W, {0: 0, 1: 3, 2: 4, 3: 1}[epi], bias=bias, resid=resid)
    out = torch.full((R, nout), float("nan"), dtype=BF, device="cuda")
    sync(lib.aigv_op_skinny_gemm(ptr(dev(x)), K, R, ptr(dev(W)), K, N, K, ptr(dev(bias)) if bias is not None else None,
                                 ptr(dev(resid)) if resid is not None else None, nout, ptr(out), nout, epi, None), lib)
    ulp_check(out, want, frac=0.03, max_ulps=4 if epi in (2, 3) else 2, atol_rel=2.0 ** -7 if epi == 1 else 2e-5)


@pytest.mark.parametrize("p", [2, 4])
@pytest.mark.parametrize("R,N,K,epi", [(1, 512, 1024, 1), (3, 1024, 4096, 2), (4, 768, 3584, 1), (8, 640, 512, 0), (2, 96, 512, 2), (1, 4096, 14336, 1)])
def test_skinny_gemm_sub_slab_forms(lib, p, R, N, K, epi):
    """The 8- and 4-row forms of the decode GEMVs (K sub-ranges packed into the MFMA's spare rows / columns): the same checks as the
    16-row form, and agreement with it up to fp32 summation order."""
    from aigv_assessor_amd.native import ptr
    if R > 16 // p:
        pytest.skip("this form takes at most 16 / p rows")
    g = torch.Generator().manual_seed(R + N + K + p)
    x = (torch.randn(R, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi == 0 else None
    nout = N // 2 if epi == 2 else N
    resid = torch.randn(R, nout, generator=g).to(BF) if epi == 1 else None
    want = gemm_ref(x, W, {0: 0, 1: 3, 2: 4}[epi], bias=bias, resid=resid)
    outs = []
    try:
        for form in (1, p):
            assert lib.aigv_tune_skinny(form) == 0
            out = torch.full((R, nout), float("nan"), dtype=BF, device="cuda")
            sync(lib.aigv_op_skinny_gemm(ptr(dev(x)), K, R, ptr(dev(W)), K, N, K, ptr(dev(bias)) if bias is not None else None,
                                         ptr(dev(resid)) if resid is not None else None, nout, ptr(out), nout, epi, None), lib)
            ulp_check(out, want, frac=0.03, max_ulps=4 if epi == 2 else 2, atol_rel=2.0 ** -7 if epi == 1 else 2e-5)
            outs.append(out.float().cpu())
    finally:
        lib.aigv_tune_skinny(0)
    assert (outs[0] != outs[1]).float().mean() < 0.03            # the two forms differ by summation order only


@pytest.mark.parametrize("R,V,H", [(10, 1009, 256), (40, 92553, 512), (64, 4099, 128)])
def test_lm_head_argmax_first_max_on_ties(lib, R, V, H):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(V)
    h = torch.randn(R, H, generator=g).to(BF)
    W = (torch.randn(V, H, generator=g) / math.sqrt(H)).to(BF)
    W[V - 1] = W[3]            # exact ties between a low and the very last vocabulary row
    W[17] = W[3]
    logits = rb(h.float() @ W.float().t())       # bf16 lm-head output, then .float() (modeling_internlm2.py:1095-1096)
    want = logits.argmax(-1)
    # torch.argmax returns the first maximal index; make the tie decisive for a few rows
    idx = torch.empty(R, dtype=torch.long, device="cuda")
    val = torch.empty(R, dtype=torch.float32, device="cuda")
    scratch = torch.zeros(64, dtype=torch.int64, device="cuda")
    sync(lib.aigv_op_lm_head_argmax(ptr(dev(h)), R, H, ptr(dev(W)), V, ptr(scratch), ptr(idx), ptr(val), None), lib)
    got = idx.cpu()
    gv = val.cpu()
    # the winning VALUE must match the rounded reference to 1 ulp; where it matches exactly the index must too
    wv = logits.max(-1).values
    same = gv == wv
    assert same.float().mean() > 0.9
    assert torch.equal(got[same], want[same])
    assert ((gv - wv).abs() <= wv.abs() * 2.0 ** -7 + 1e-6).all()
    h2 = W[3:4].clone().expand(3, H).contiguous()   # rows whose best vocab rows are the tied 3/17/V-1 -> must return 3
    sync(lib.aigv_op_lm_head_argmax(ptr(dev(h2)), 3, H, ptr(dev(W)), V, ptr(scratch), ptr(idx), ptr(val), None), lib)
    l2 = rb(h2.float() @ W.float().t())
    assert torch.equal(idx.cpu()[:3], l2.argmax(-1))


def test_frame_ingest_is_bit_exact(lib):
    """uint8 HWC -> bf16 NCHW normalisation: three fp32 ops and one bf16 rounding -> identical to the oracle."""
    import ctypes as C
    from aigv_assessor_amd.native import ptr
    from oracle import oracle as O
    g = torch.Generator().manual_seed(5)
    u = torch.randint(0, 256, (3, 28, 44, 3), generator=g, dtype=torch.uint8)
    u[0, 0, 0] = torch.tensor([0, 255, 128], dtype=torch.uint8)
    want = O.normalize_frames_u8(u)
    out = torch.empty(3, 3, 28, 44, dtype=BF, device="cuda")
    mean, std = (C.c_float * 3)(*O.IMAGENET_MEAN), (C.c_float * 3)(*O.IMAGENET_STD)
    sync(lib.aigv_op_frame_ingest(ptr(dev(u)), 3, 28, 44, mean, std, ptr(out), None), lib)
    assert torch.equal(out.cpu(), want)


# ---------------------------------------------------------------------------------------------------------
# frame resize + ingest: Pillow-exact BICUBIC on the GPU
# ---------------------------------------------------------------------------------------------------------
def _resize_ingest(lib, frames_u8, oh, ow, want_u8=True, want_nchw=True):
    import ctypes
    from aigv_assessor_amd.native import ptr
    f = dev(torch.from_numpy(frames_u8))
    n, ih, iw, _ = f.shape
    tmp = torch.empty(n * ih * ow * 3, dtype=torch.uint8, device="cuda")
    u8 = torch.full((n, oh, ow, 3), 77, dtype=torch.uint8, device="cuda") if want_u8 else None
    pv = torch.full((n, 3, oh, ow), float("nan"), dtype=BF, device="cuda") if want_nchw else None
    mean, std = (ctypes.c_float * 3)(0.485, 0.456, 0.406), (ctypes.c_float * 3)(0.229, 0.224, 0.225)
    sync(lib.aigv_op_frame_resize_ingest(ptr(f), n, ih, iw, oh, ow, mean, std, ptr(tmp), ptr(u8), ptr(pv), None), lib)
    return u8, pv


def test_frame_resize_is_bit_exact_with_pillow_fixtures(lib):
    """The recorded Pillow results (tests/golden/resize.npz): every byte equal; the fused bf16 NCHW output equals the
    ToTensor / Normalize / bf16 restatement applied to Pillow's bytes."""
    import os
    import numpy as np
    from oracle import oracle as O
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize.npz"))
    n = len([k for k in z.files if k.startswith("in")])
    for i in range(n):
        want = z[f"out{i}"]
        u8, pv = _resize_ingest(lib, z[f"in{i}"][None], want.shape[0], want.shape[1])
        assert np.array_equal(u8.cpu().numpy()[0], want), i
        ref = O.normalize_frames_u8(torch.from_numpy(want)[None])
        assert torch.equal(pv.cpu(), ref), i


@pytest.mark.parametrize("ih,iw,oh,ow,frames", [(720, 1280, 448, 448, 3), (1080, 1920, 448, 448, 2), (224, 224, 448, 448, 4),
                                                (448, 448, 448, 448, 2), (360, 640, 224, 224, 5), (37, 1000, 20, 30, 1)])
def test_frame_resize_matches_oracle_at_video_sizes(lib, ih, iw, oh, ow, frames):
    """Seeded frames at real video resolutions (several frames per call: the frame stride of both passes) against the CPU
    restatement of Pillow's algorithm - bit-exact, uint8 and normalised bf16 outputs, each also on its own."""
    import numpy as np
    from oracle import oracle as O
    from oracle.resize import resize_bicubic_u8
    rng = np.random.default_rng(ih * 7 + iw + frames)
    fr = rng.integers(0, 256, (frames, ih, iw, 3), dtype=np.uint8)
    fr[0, : ih // 2] = np.where(rng.random((ih // 2, iw, 3)) < 0.5, 0, 255)      # hard edges in one frame
    want = np.stack([resize_bicubic_u8(fr[i], oh, ow) for i in range(frames)])
    u8, pv = _resize_ingest(lib, fr, oh, ow)
    assert np.array_equal(u8.cpu().numpy(), want)
    ref = O.normalize_frames_u8(torch.from_numpy(want))
    assert torch.equal(pv.cpu(), ref)
    u8_only, none = _resize_ingest(lib, fr, oh, ow, want_nchw=False)
    assert none is None and torch.equal(u8_only, u8)
    none, pv_only = _resize_ingest(lib, fr, oh, ow, want_u8=False)
    assert none is None and torch.equal(pv_only, pv)


def test_frame_resize_rejects_bad_arguments(lib):
    from aigv_assessor_amd import native
    t = torch.zeros(16, dtype=torch.uint8, device="cuda")
    assert lib.aigv_op_frame_resize_ingest(t.data_ptr(), 1, 2, 2, 2, 2, None, None, t.data_ptr(), None, None, None) != 0   # no output
    assert lib.aigv_op_frame_resize_ingest(t.data_ptr(), 1, 0, 2, 2, 2, None, None, t.data_ptr(), t.data_ptr(), None, None) != 0
    assert lib.aigv_op_frame_resize_ingest(t.data_ptr(), 1, 2, 2, 2, 2, None, None, t.data_ptr(), None, t.data_ptr(), None) != 0   # nchw needs mean/std
    # frames more than 100 times taller than wide that shrink vertically: Pillow runs its vertical pass first there (tests/manual/fuzz_resize.py) - refused
    src, tmp, out = (torch.zeros(n, dtype=torch.uint8, device="cuda") for n in (3 * 801 * 8, 3 * 801 * 448, 3 * 448 * 448))
    assert lib.aigv_op_frame_resize_ingest(src.data_ptr(), 1, 801, 8, 448, 448, None, None, tmp.data_ptr(), out.data_ptr(), None, None) != 0
    assert b"100 times taller" in lib.aigv_last_error(None)
    assert lib.aigv_op_frame_resize_ingest(src.data_ptr(), 1, 800, 8, 448, 448, None, None, tmp.data_ptr(), out.data_ptr(), None, None) == 0
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------------------
# fp8 groundwork (BASELINE config 5): row quantisation and the e4m3 GEMM, each against a torch restatement of its arithmetic
# ---------------------------------------------------------------------------------------------------------
def _quant_ref(x_bf16):
    x = x_bf16.float()
    amax = x.abs().amax(dim=-1, keepdim=True)
    inv = torch.where(amax > 0, torch.full_like(amax, 448.0) / amax, torch.ones_like(amax))   # a true IEEE division (scalar / tensor is rcp * scalar)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (x * inv).to(torch.float8_e4m3fn), scale.reshape(-1)


@pytest.mark.parametrize("rows,K", [(7, 128), (300, 4096), (33, 14336), (5, 1032)])
def test_fp8_row_quantisation_is_bit_exact(lib, rows, K):
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * torch.rand(rows, 1, generator=g) * 8).to(BF)
    x[0, : K // 2] = 0
    if rows > 3:
        x[3] = 0                                                    # an all-zero row: scale 1, bytes 0
        x[2, 5] = 300.0                                             # an outlier: everything else lands in the low binades
    q_ref, s_ref = _quant_ref(x)
    q = torch.full((rows, K), 0x55, dtype=torch.uint8, device="cuda")
    sc = torch.full((rows,), -1.0, dtype=torch.float32, device="cuda")
    sync(lib.aigv_op_quant_fp8_rows(ptr(dev(x)), K, rows, K, ptr(q), K, ptr(sc), None), lib)
    assert torch.equal(sc.cpu(), s_ref)
    got, want = q.cpu(), q_ref.view(torch.uint8)
    same = got == want
    zero_sign = ((got & 0x7f) == 0) & ((want & 0x7f) == 0)          # +0 / -0 are the same value
    assert bool((same | zero_sign).all()), int((~(same | zero_sign)).sum())


@pytest.mark.parametrize("M,N,K,bias", [(256, 256, 128, False), (300, 512, 1024, True), (1029, 768, 384, False), (2048, 1024, 4096, True)])
def test_fp8_gemm_matches_its_arithmetic(lib, M, N, K, bias):
    """e4m3 x e4m3 products are exact in fp32 and the MFMA accumulates in fp32: against the dequantised fp32 matmul, scaled and rounded
    to bf16 the same way, only the summation order differs -> at most 1 bf16 ulp on a small fraction of the elements."""
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.7).to(BF)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    b = (torch.randn(N, generator=g) * 0.1).to(BF) if bias else None
    qa, sa = _quant_ref(a)
    qw, sw = _quant_ref(w)
    acc = qa.float().double() @ qw.float().double().t()            # exact products, fp64 sum
    ref = (acc.float() * sa[:, None]) * sw[None, :]
    if bias:
        ref = ref + b.float()
    want = rb(ref)
    dA, dW = dev(qa.view(torch.uint8)), dev(qw.view(torch.uint8))
    dsa, dsw = dev(sa), dev(sw)
    C = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
    sync(lib.aigv_op_gemm_fp8(ptr(dA), K, ptr(dW), K, ptr(C), N, ptr(dsa), ptr(dsw), ptr(dev(b)) if bias else None, None, None, 0, M, N, K, 0, 0, None, None), lib)
    ulp_check(C, want, frac=0.02, max_ulps=1, atol_rel=2e-5)
    # and through the quantisation kernel end to end: the same bytes come out of aigv_op_quant_fp8_rows
    q = torch.empty((M, K), dtype=torch.uint8, device="cuda"); sc = torch.empty(M, dtype=torch.float32, device="cuda")
    sync(lib.aigv_op_quant_fp8_rows(ptr(dev(a)), K, M, K, ptr(q), K, ptr(sc), None), lib)
    C2 = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
    sync(lib.aigv_op_gemm_fp8(ptr(q), K, ptr(dW), K, ptr(C2), N, ptr(sc), ptr(dsw), ptr(dev(b)) if bias else None, None, None, 0, M, N, K, 0, 0, None, None), lib)
    assert torch.equal(C2, C)


def _epilogue_ref(acc, epi, bias=None, ls=None, resid=None):
    """gemm_ref's rounding chain applied to a given fp32 accumulator."""
    if epi == 4:
        blk = acc.view(acc.shape[0], acc.shape[1] // 32, 2, 16)
        g, u = rb(blk[:, :, 0, :]).reshape(acc.shape[0], -1), rb(blk[:, :, 1, :]).reshape(acc.shape[0], -1)
        return rb(rb(torch.nn.functional.silu(g)) * u)
    if bias is not None:
        acc = acc + bias.float()
    y = rb(acc)
    if epi == 1:
        y = rb(torch.nn.functional.gelu(y))
    if epi == 2:
        y = rb(y * ls.float())
    if epi in (2, 3):
        y = rb(resid.float() + y)
    return y


@pytest.mark.parametrize("epi", [1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (1029, 768, 1024)])
def test_fp8_gemm_epilogues(lib, M, N, K, epi):
    """The fp8 form keeps every rounding point of the bf16 kernel's epilogues; its scaled accumulator replaces the bf16 accumulator."""
    from aigv_assessor_amd.native import ptr
    g = torch.Generator().manual_seed(M + N + K + epi)
    a = (torch.randn(M, K, generator=g) * 0.7).to(BF)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF) if epi in (1, 2) else None
    ls = (torch.rand(N, generator=g) + 0.5).to(BF) if epi == 2 else None
    nout = N // 2 if epi == 4 else N
    resid = torch.randn(M, nout, generator=g).to(BF) if epi in (2, 3) else None
    qa, sa = _quant_ref(a)
    qw, sw = _quant_ref(w)
    acc = ((qa.float().double() @ qw.float().double().t()).float() * sa[:, None]) * sw[None, :]
    want = _epilogue_ref(acc, epi, bias, ls, resid)
    C = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
    db, dl, dr = (dev(t) if t is not None else None for t in (bias, ls, resid))
    sync(lib.aigv_op_gemm_fp8(ptr(dev(qa.view(torch.uint8))), K, ptr(dev(qw.view(torch.uint8))), K, ptr(C), nout, ptr(dev(sa)), ptr(dev(sw)),
                              ptr(db), ptr(dl), ptr(dr), nout, M, N, K, epi, 0, None, None), lib)
    ulp_check(C, want, frac=0.03 if epi in (1, 4) else 0.02, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)
    # split-K form (scaled fp32 slabs + the bf16 path's finalize pass): same result up to the summation order
    ws = torch.empty(2 * M * N, dtype=torch.float32, device="cuda")
    C2 = torch.full((M, nout), float("nan"), dtype=BF, device="cuda")
    sync(lib.aigv_op_gemm_fp8(ptr(dev(qa.view(torch.uint8))), K, ptr(dev(qw.view(torch.uint8))), K, ptr(C2), nout, ptr(dev(sa)), ptr(dev(sw)),
                              ptr(db), ptr(dl), ptr(dr), nout, M, N, K, epi, 2, ptr(ws), None), lib)
    ulp_check(C2, want, frac=0.03 if epi in (1, 4) else 0.02, max_ulps=4 if epi in (1, 4) else 2, atol_rel=2.0 ** -7 if epi in (2, 3) else 2e-5)


@pytest.mark.parametrize("p", [1, 2, 4])
@pytest.mark.parametrize("R,N,K,epi,norm", [(1, 512, 4096, 1, False), (3, 256, 14336, 1, False), (4, 1024, 4096, 2, True), (2, 384, 6144, 2, True),
                                            (1, 128, 16384, 1, False)])
def test_fp8_decode_gemv(lib, p, R, N, K, epi, norm):
    """The e4m3 form of the decode GEMVs (csrc/head8.hip): the kernel normalises (optionally) and quantises the x rows itself; the
    result is the fp8 mode's definition (oracle/fp8.py: per-row / per-channel scales, exact products, fp32 sums) with the bf16
    form's epilogue rounding points."""
    from aigv_assessor_amd.native import ptr
    if R > 16 // p:
        pytest.skip("this form takes at most 16 / p rows")
    g = torch.Generator().manual_seed(R + N + K + p)
    x = (torch.randn(R, K, generator=g) * 0.7).to(BF)
    x[:, 5] *= 6.0                                                 # an outlier per row: the amax is not a typical element
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(BF)
    gw = (torch.rand(K, generator=g) + 0.5).to(BF) if norm else None
    eps = 1e-5
    nout = N // 2 if epi == 2 else N
    resid = torch.randn(R, nout, generator=g).to(BF) if epi == 1 else None
    xin = x
    if norm:                                                       # InternLM2RMSNorm: fp32 normalise -> bf16 -> * weight -> bf16
        xf = x.float()
        xin = (gw.float() * rb(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))).to(BF)
    qa, sa = _quant_ref(xin)
    qw, sw = _quant_ref(w)
    acc = ((qa.float().double() @ qw.float().double().t()).float() * sa[:, None]) * sw[None, :]
    want = _epilogue_ref(acc, 4 if epi == 2 else 3, None, None, resid)
    out = torch.full((R, nout), float("nan"), dtype=BF, device="cuda")
    dsw = dev(sw)
    import ctypes as C
    sync(lib.aigv_op_skinny_gemm_fp8(ptr(dev(x)), K, R, ptr(dev(qw.view(torch.uint8))), K, C.cast(dsw.data_ptr(), C.POINTER(C.c_float)), N, K,
                                     ptr(dev(resid)) if resid is not None else None, nout, ptr(out), nout, epi, ptr(dev(gw)) if norm else None, eps, p, None), lib)
    ulp_check(out, want, frac=0.03, max_ulps=4 if epi == 2 else 2, atol_rel=2.0 ** -7 if epi == 1 else 2e-5)


def test_fp8_gemm_rejects_bad_shapes(lib):
    t = torch.zeros(256 * 256, dtype=torch.uint8, device="cuda")
    f = torch.ones(256, dtype=torch.float32, device="cuda")
    c = torch.empty(256, 256, dtype=BF, device="cuda")
    ok = lambda **kw: lib.aigv_op_gemm_fp8(t.data_ptr(), kw.get("K", 128), t.data_ptr(), kw.get("K", 128), c.data_ptr(), 256, kw.get("sa", f.data_ptr()),
                                           f.data_ptr(), None, None, None, 0, 256, kw.get("N", 256), kw.get("K", 128), 0, 0, None, None)
    assert ok() == 0
    assert ok(N=128) != 0 and ok(K=64) != 0 and ok(sa=None) != 0
    torch.cuda.synchronize()
