"""SlowFast-R50 motion branch on the GPU, through the C ABI, against oracle/slowfast.py.

PARITY UNPINNED against pytorchvideo itself (absent offline); what is pinned here is the HIP branch against the CPU
restatement of the published architecture, and the convolution kernel against torch's conv3d."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def lib():
    from aigv_assessor_amd import native
    return native.load()


def _pack_conv(w):
    """[Cout, Cin, kt, kh, kw] fp32 -> kernel layout [ceil16(Cout), Kp] bf16, k = tap * Cin + ci."""
    cout, cin = w.shape[:2]
    k = w.permute(0, 2, 3, 4, 1).reshape(cout, -1)
    kp = (k.shape[1] + 63) // 64 * 64
    out = torch.zeros((cout + 15) // 16 * 16, kp)
    out[:cout, : k.shape[1]] = k
    return out.to(BF), kp


CASES = [  # B, Cin, Ti, Hi, Wi, Cout, (kt,kh,kw), (st,sh,sw), (pt,ph,pw), residual, relu, extra ld / c_off
    (2, 8, 4, 20, 20, 8, (5, 7, 7), (1, 2, 2), (2, 3, 3), False, True, 0),       # the fast stem's geometry (Cin padded to 8)
    (1, 80, 2, 14, 14, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, True, 0),     # conv_a, Cin not a power of two
    (2, 16, 8, 12, 12, 16, (3, 1, 1), (1, 1, 1), (1, 0, 0), False, True, 0),     # temporal conv_a
    (2, 64, 2, 15, 13, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1), False, True, 0),     # strided conv_b, odd map
    (1, 32, 2, 9, 9, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, True, 0),       # conv_c with residual
    (1, 24, 3, 9, 9, 40, (1, 1, 1), (1, 2, 2), (0, 0, 0), False, False, 0),      # shortcut: strided 1x1x1, no ReLU, Cout % 16 != 0
    (2, 32, 8, 6, 6, 64, (7, 1, 1), (4, 1, 1), (3, 0, 0), False, True, 24),      # fast->slow fusion written at a channel offset
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_kernel_matches_torch(lib, case):
    from aigv_assessor_amd.native import check, ptr
    B, cin, Ti, Hi, Wi, cout, k, s, p, use_res, relu, c_off = case
    g = torch.Generator().manual_seed(cin * 131 + cout)
    x = torch.randn(B, Ti, Hi, Wi, cin + 8, generator=g).to(BF)                  # ld_in = Cin + 8: the kernel must ignore the tail channels
    w = (torch.randn(cout, cin, *k, generator=g) / math.sqrt(cin * k[0] * k[1] * k[2])).to(BF).float()
    bias = torch.randn(cout, generator=g) * 0.2
    ref = F.conv3d(x[..., :cin].float().permute(0, 4, 1, 2, 3), w, bias, s, p).permute(0, 2, 3, 4, 1)   # channels-last
    To, Ho, Wo = ref.shape[1:4]
    res = torch.randn(B, To, Ho, Wo, cout, generator=g).to(BF) if use_res else None
    if use_res:
        ref = ref + res.float()
    if relu:
        ref = ref.relu()
    wp, kp = _pack_conv(w)
    ld_out = cout + c_off + 4
    out = torch.full((B, To, Ho, Wo, ld_out), 7.0, dtype=BF, device="cuda")
    dims = (C.c_int * 12)(Ti, Hi, Wi, *k, *s, *p)
    xd, wd, bd = x.cuda(), wp.cuda(), bias.cuda()
    rd = res.cuda() if use_res else None
    check(lib.aigv_op_conv3d(ptr(xd), cin + 8, cin, B, dims, ptr(wd), kp, ptr(bd), cout, ptr(rd), cout, ptr(out), ld_out, c_off, int(relu), None))
    torch.cuda.synchronize()
    got = out.cpu().float()
    assert torch.all(got[..., :c_off] == 7.0) and torch.all(got[..., c_off + cout:] == 7.0)          # nothing outside the channel window
    d = (got[..., c_off:c_off + cout] - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2e-3                                                                # bf16 output rounding + fp32 summation order
    assert bool((d <= tol).all()), float((d - tol).max())


def test_conv3d_rejects_misaligned(lib):
    t = torch.zeros(4096, dtype=BF, device="cuda")
    f = torch.zeros(64, dtype=torch.float32, device="cuda")
    dims = (C.c_int * 12)(1, 4, 4, 1, 1, 1, 1, 1, 1, 0, 0, 0)
    call = lambda cin=8, cout=16, kp=64, ld_out=16: lib.aigv_op_conv3d(t.data_ptr(), cin, cin, 1, dims, t.data_ptr(), kp, f.data_ptr(), cout, None, 0, t.data_ptr(), ld_out, 0, 1, None)
    assert call() == 0
    assert call(cin=4) != 0 and call(cout=6, ld_out=6) != 0 and call(kp=32) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,T,S", [(2, 8, 224), (1, 16, 256), (1, 12, 224)])   # T = 12: slow frames 0, 5, 11; 5 temporal windows in the slow pool
def test_slowfast_branch_matches_oracle(B, T, S):
    from aigv_assessor_amd import synth
    from aigv_assessor_amd.slowfast import SlowFastR50
    from oracle import slowfast as osf
    sd = synth.slowfast_state_dict(seed=3)
    g = torch.Generator().manual_seed(B * 100 + T)
    frames = torch.randn(B * T, 3, S, S, generator=g).clamp(-2.5, 2.5).to(BF)
    clip = frames.view(B, T, 3, S, S).permute(0, 2, 1, 3, 4)
    want32 = osf.slowfast_features(sd, clip.float())                      # fp32 arithmetic on the same bf16 frames
    want16 = osf.slowfast_features(sd, clip).float()                      # the reference's dtype flow (bf16 modules)
    sf = SlowFastR50(sd)
    got = sf.features(frames.cuda(), B).float().cpu()
    assert got.shape == (B, 2304)
    ref_err = (want16 - want32).abs()
    err = (got - want32).abs()
    scale = want32.abs().mean()
    # ~110 bf16-rounded layers: the HIP branch (folded norms, fp32 accumulation, one rounding per conv) must sit at least as close to
    # fp32 as the bf16 module flow does, up to a small constant
    assert err.mean() <= 1.5 * ref_err.mean() + 1e-3 * scale, (float(err.mean()), float(ref_err.mean()))
    assert err.max() <= 2.0 * ref_err.max() + 5e-3 * scale, (float(err.max()), float(ref_err.max()))
    # reference call form: [slow, fast] -> [B, 2304, 1, 1, 1]
    fast = clip.cuda()
    again = sf([fast.index_select(2, torch.linspace(0, T - 1, T // 4).long().cuda()), fast])
    assert again.shape == (B, 2304, 1, 1, 1) and torch.equal(again.view(B, -1).float().cpu(), got)
    # clips are independent: one clip alone gives the same bits
    one = sf.features(frames[:T].cuda(), 1).float().cpu()
    assert torch.equal(one[0], got[0])


def test_slowfast_reports_missing_weights():
    from aigv_assessor_amd import native, synth
    from aigv_assessor_amd.slowfast import SlowFastR50
    sd = synth.slowfast_state_dict(seed=0)
    del sd[synth.SLOWFAST_PREFIX + "3.multipathway_blocks.1.res_blocks.2.branch2.norm_b.running_var"]
    sf = SlowFastR50(sd)
    with pytest.raises(native.NativeError, match="res_blocks.2.branch2.norm_b.running_var"):
        sf.features(torch.zeros(8, 3, 224, 224, dtype=BF, device="cuda"), 1)


def test_model_uses_the_native_branch_when_the_checkpoint_carries_it():
    """stage-2 forward without motion_feature: the state dict's slowfast_model.* tensors feed the native branch
    (modeling_internvl_chat.py:336-345); the result equals passing that branch's feature explicitly, and the oracle's scorer fed with the
    ORACLE's SlowFast feature agrees on the level tokens and the score (the whole row E -> I chain, both sides independent)."""
    import aigv_assessor_amd as pkg
    from aigv_assessor_amd import synth
    from aigv_assessor_amd.modeling import InternVLChatModel
    from oracle import oracle as O
    from oracle import slowfast as osf
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    B, T = 2, 8
    sd = synth.make_state_dict(cfg, seed=5, rich=True)
    sf_sd = synth.slowfast_state_dict(seed=5)
    toks = synth.canonical_tokens(cfg, B, T, seed=5)
    pv = synth.synthetic_frames(B * T, 224, seed=5)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    model = InternVLChatModel(cfg, stage=2)
    model.load_state_dict({**sd, **sf_sd})
    model = model.eval().cuda()
    model.img_context_token_id = toks["img_context_token_id"]
    kw = dict(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"])
    out = model(**kw)
    feat = model.slowfast_model.features(pv.cuda(), B)
    out2 = model(motion_feature=feat, **kw)
    assert torch.equal(out["score1"], out2["score1"]) and torch.equal(out["logit"], out2["logit"])
    motion = osf.slowfast_features(sf_sd, pv.view(B, T, 3, 224, 224).permute(0, 2, 1, 3, 4))          # bf16 module flow, CPU
    ref = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, toks["img_context_token_id"],
                         mos=None, stage=2)
    want = ref["label"] != -100
    assert int((out["logit"].cpu()[want] != ref["logit"][want]).sum()) <= 1
    d = (out["score1"].float().cpu() - ref["score1"].float()).abs()
    assert bool((d <= 2.0 ** -6 * ref["score1"].float().abs().clamp_min(1.0) * 1.001).all()), d
