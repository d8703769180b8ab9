"""End-to-end parity (MI355X only): the product path (InternVLChatModel -> C ABI -> HIP kernels) against the CPU
oracle on the same seeded inputs, and against the golden vectors recorded from the reference itself.

Bars (BASELINE.json north_star): quality-level tokens (answer-row argmax ids) bit-exact; score1 within 1e-3.
score1 is a bf16 number in the reference (one ulp = 3.9e-3 in [0.5, 1), 2e-3 in [0.25, 0.5)), and the reference's own
bf16 path sits up to 6e-3 away from its fp32 path at 2 layers and 0.002-0.026 at full depth (profiles/parity_score_noise_r2.txt,
profiles/parity_full_size_r2.txt), so "within 1e-3" is only reachable as "the same bf16 value".  Tolerances written in the tests:
  * tiny configurations: |d| <= 1e-3 OR <= 1 bf16 ulp of the expected value (score_ok);
  * 4096-wide shallow configurations: <= 4 bf16 ulps AND anchored on the fp32 oracle (score_near_fp32);
  * full depth (32 + 24 layers) against the REFERENCE's recorded outputs: per clip <= 1.3 x the largest distance of the reference's bf16
    pass to ITSELF under another host thread count, pooled mean over 37 clips <= 1.3 x its mean distance (both read from
    tests/golden/e2e_8b_r5.pt: 8.0 / 2.56 bf16 ulps over 44 pairs), and over the clips as close to its fp32 value as its own bf16 pass is
    (test_full_size_8b_* below);
  * level tokens: identical, except rows where the reference's OWN top logits are within 2 (tiny) / 4 (full depth) bf16 ulps; hard
    equality on the planted-margin weights.
"""
import os

import pytest
import torch

import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def make_model(cfg, sd, stage=2, **kw):
    from aigv_assessor_amd.modeling import InternVLChatModel
    m = InternVLChatModel(cfg, stage=stage, **kw)
    m.load_state_dict(sd)
    return m.eval().cuda()


def rel_close(got, want, mean_tol, max_tol):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    scale = want.abs().mean().clamp_min(1e-6)
    err = (got - want).abs()
    assert err.mean() / scale <= mean_tol, f"mean rel err {(err.mean() / scale).item():.4g}"
    assert err.max() / scale <= max_tol, f"max rel err {(err.max() / scale).item():.4g}"


SCORE_LOG = []   # (test id, max |d| in bf16 ulps of the expected value) - printed so that a run shows the margin under each bar


def score_ok(got, want, ulps=1):
    """`score1` against the oracle / the reference's recorded value.  The reference emits a bf16 number, so the bar is stated in bf16
    ulps of the expected value, per model size as measured (profiles/parity_score_noise_r1.txt): 1 ulp for the tiny
    configurations, 2 for the 4096-wide shallow ones; 1e-3 (BASELINE's figure) always passes.  Full depth has its own statistical bar
    (test_full_size_8b_matches_the_reference_golden)."""
    got, want = got.float().cpu(), want.float().cpu()
    d = (got - want).abs()
    ulp = want.abs().clamp_min(2.0 ** -126).log2().floor().exp2() * 2.0 ** -7
    n_ulps = (d / ulp).max().item()
    SCORE_LOG.append(n_ulps)
    print("score1 hip", got.tolist(), "oracle", want.tolist(), "max|d|", d.max().item(), f"= {n_ulps:.2f} bf16 ulps (bar {ulps})")
    assert bool(((d <= 1e-3) | (d <= ulps * ulp * 1.001)).all()), f"score differs by {n_ulps:.2f} bf16 ulps (bar {ulps}): {got.tolist()} vs {want.tolist()}"


def score_near_fp32(got, ref_bf16, cfg, sd, toks, pv, motion, flags):
    """The 4096-wide shallow configurations: a clip's score can sit 3-4 bf16 ulps from the reference's bf16 value while being CLOSER to
    the fp32 computation than that value is (measured on tests/golden/e2e.pt bf16_b2: fp32 0.7202, reference bf16 0.7305, HIP 0.7148;
    the HIP value itself moves by +-2 ulps with the attention / GEMM kernel choice).  So besides the ulp bar against the bf16 value the
    score is anchored on the fp32 oracle, per clip: |hip - fp32| <= 1.5 |reference bf16 - fp32| + 1 bf16 ulp, or within 4 bf16 ulps of the
    fp32 value (the largest single-clip deviations seen at this width: 3.8 ulps here, 4.6 in round 1's study; over 12 seeds the MEAN
    |hip - fp32| is 0.0021 against 0.0027 for the reference's bf16 path: profiles/parity_score_noise_r2.txt)."""
    sd32 = {k: v.float() for k, v in sd.items()}
    f32 = O.forward_eval(sd32, cfg, pv.float(), toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion.float(),
                         toks["img_context_token_id"], stage=2)["score1"].float()
    got, ref_bf16 = got.float().cpu(), ref_bf16.float().cpu()
    ulp = f32.abs().clamp_min(2.0 ** -126).log2().floor().exp2() * 2.0 ** -7
    print("score1 fp32 oracle", f32.tolist(), "|hip - fp32|", (got - f32).abs().tolist(), "|reference bf16 - fp32|", (ref_bf16 - f32).abs().tolist())
    d = (got - f32).abs()
    assert bool(((d <= 1.5 * (ref_bf16 - f32).abs() + ulp) | (d <= 4 * ulp * 1.001)).all())


def run_case(cfg, B, T, seed, stage=2, px=None):
    px = px or cfg.image_size
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    pv = synth.synthetic_frames(B * T, px, seed=seed)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    mos = torch.full((B,), 0.5, dtype=BF)
    ref = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                         toks["img_context_token_id"], mos=mos, stage=stage, return_intermediates=True)
    model = make_model(cfg, sd, stage=stage)
    model.img_context_token_id = toks["img_context_token_id"]
    out = model(mos=mos, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                image_flags=flags, labels=toks["labels"], motion_feature=motion)
    torch.cuda.synchronize()
    return model, sd, toks, pv, motion, ref, out


def assert_levels(got_ids, want_ids, ref_logits_rows):
    """Quality-level tokens must be identical.  The one admissible exception: a row where the REFERENCE's own
    logits for the two candidate tokens are within 2 bf16 ulps of each other (a tie inside its own rounding noise —
    only seen with random-weight models, whose vocabulary logits are near-uniform).  Every such row is printed."""
    got_ids, want_ids = got_ids.cpu(), want_ids.cpu()
    bad = (got_ids != want_ids).nonzero().flatten().tolist()
    for r in bad:
        a, b = ref_logits_rows[r, want_ids[r]].item(), ref_logits_rows[r, got_ids[r]].item()
        ulp = 2.0 ** (torch.tensor(abs(a)).clamp_min(1e-30).log2().floor().item() - 7)
        print(f"level row {r}: oracle id {want_ids[r].item()} ({a}) vs hip id {got_ids[r].item()} ({b}); gap {abs(a - b) / ulp:.2f} ulp")
        assert abs(a - b) <= 2 * ulp, f"row {r}: argmax differs beyond a rounding tie"
    return len(bad)


def assert_levels_recorded(got_ids, want_ids, top_ids, top_values):
    """assert_levels with the REFERENCE's own recorded logits (top 8 per answer row: tests/golden/e2e_anchors.pt) in place of a live
    oracle pass: a differing HIP token must be among the reference's recorded candidates, within 2 bf16 ulps of its winner."""
    got_ids, want_ids = got_ids.cpu(), want_ids.cpu()
    assert torch.equal(top_ids[:, 0], want_ids)
    bad = (got_ids != want_ids).nonzero().flatten().tolist()
    for r in bad:
        ids = top_ids[r].tolist()
        assert int(got_ids[r]) in ids, f"row {r}: hip id {int(got_ids[r])} is not among the reference's top 8 {ids}"
        a, b = top_values[r, 0].item(), top_values[r, ids.index(int(got_ids[r]))].item()
        ulp = 2.0 ** (torch.tensor(abs(a)).clamp_min(1e-30).log2().floor().item() - 7)
        print(f"level row {r}: reference id {want_ids[r].item()} ({a}) vs hip id {got_ids[r].item()} ({b}); gap {abs(a - b) / ulp:.2f} ulp")
        assert abs(a - b) <= 2 * ulp, f"row {r}: argmax differs beyond a rounding tie"
    return len(bad)


def score_near_recorded_fp32(got, ref_bf16, f32):
    """score_near_fp32 with the fp32 anchor recorded from the reference (its fp32 pass over the bf16-rounded weights and inputs:
    tests/golden/make_golden_anchors.py) instead of a live fp32 oracle pass; same bar."""
    got, ref_bf16, f32 = got.float().cpu(), ref_bf16.float().cpu(), f32.float().cpu()
    ulp = f32.abs().clamp_min(2.0 ** -126).log2().floor().exp2() * 2.0 ** -7
    print("score1 reference fp32", f32.tolist(), "|hip - fp32|", (got - f32).abs().tolist(), "|reference bf16 - fp32|", (ref_bf16 - f32).abs().tolist())
    d = (got - f32).abs()
    assert bool(((d <= 1.5 * (ref_bf16 - f32).abs() + ulp) | (d <= 4 * ulp * 1.001)).all())


def check_levels(out, ref):
    want = ref["label"] != -100
    got = out["logit"].cpu()
    assert torch.equal(out["label"].cpu(), ref["label"])
    assert (got[~want] == -1).all()
    logits = ref["logits"][..., :-1, :].reshape(-1, ref["logits"].shape[-1])[want]
    n_tie = assert_levels(got[want], ref["logit"][want], logits)
    assert n_tie <= max(1, int(want.sum()) // 10)


def test_stage2_tiny_224(capsys):
    cfg = pkg.tiny(image_size=224)
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=2, T=4, seed=7)
    # component parity on the way: pre-projector tokens, projected tokens, motion token
    vt = model.vit_tokens(pv)
    rel_close(vt, O.shuffled_tokens(O.vit_forward(sd, cfg, pv)), 0.01, 0.25)
    rel_close(model.project(vt).reshape(-1, cfg.llm_config.hidden_size), ref["vit_embeds"].reshape(-1, cfg.llm_config.hidden_size), 0.02, 0.5)
    rel_close(model.motion_embed(motion), ref["motion_embeds"], 0.01, 0.1)
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])
    assert abs(out["loss"].float().item() - ref["loss"].float().item()) <= 4e-3


@pytest.mark.parametrize("select_layer", [-2, 1])
def test_select_layer_other_than_the_last(select_layer):
    """vision_select_layer != -1 (modeling_internvl_chat.py:509-518: ``hidden_states[select_layer]`` of the InternViT encoder, whose
    tuple is [embeddings, layer 1 output, ..., layer L output], modeling_intern_vit.py:250-294): -2 = the output of the second-to-last
    layer, 1 = the output of the first.  The native ViT pass then stops after that many layers (aigv_vit_forward)."""
    cfg = pkg.tiny(image_size=224, vit_layers=3, llm_layers=1)
    cfg.select_layer = select_layer
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=2, T=2, seed=9)
    assert model.select_layer == select_layer
    vt = model.vit_tokens(pv)
    want = O.shuffled_tokens(O.vit_forward(sd, cfg, pv, select_layer))
    rel_close(vt, want, 0.01, 0.25)
    last = O.shuffled_tokens(O.vit_forward(sd, cfg, pv, -1))
    assert (want.float() - last.float()).abs().mean() > 0.05 * last.float().abs().mean()      # the layers that are skipped do matter
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])


def test_stage2_448_cls_tail_and_batch_invariance():
    cfg = pkg.tiny(image_size=448, vit_layers=1, llm_layers=1)
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=2, T=2, seed=8)
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])
    # size-independent property: scoring clips one by one gives bit-identical results to the batch
    flags1 = torch.ones(2, 1, dtype=torch.long)
    for b in range(2):
        o1 = model(mos=None, pixel_values=pv[2 * b:2 * b + 2], input_ids=toks["input_ids"][b:b + 1],
                   attention_mask=toks["attention_mask"][b:b + 1], image_flags=flags1, labels=toks["labels"][b:b + 1],
                   motion_feature=motion[b:b + 1])
        n1 = toks["input_ids"].shape[1] - 1
        assert torch.equal(o1["logit"], out["logit"][b * n1:(b + 1) * n1])
        assert torch.equal(o1["score1"], out["score1"][b:b + 1])


def test_last_layer_row_trimming_is_output_equivalent():
    """The trimmed last layer (only the consumed rows are finished, on the skinny kernel) returns what the full pass
    returns: identical level tokens, scores within one bf16 ulp (fp32 summation order of a different kernel)."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=3, T=2, seed=21)
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])
    model.set_row_trimming(False)
    try:
        full = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                     image_flags=torch.ones(pv.shape[0], 1, dtype=torch.long), labels=toks["labels"], motion_feature=motion)
    finally:
        model.set_row_trimming(True)
    check_levels(full, ref)
    want = ref["label"] != -100
    n_diff = int((full["logit"].cpu()[want] != out["logit"].cpu()[want]).sum())
    assert n_diff <= 1, n_diff                      # a tie inside rounding noise may flip between the two kernels
    d = (full["score1"].float() - out["score1"].float()).abs().cpu()
    assert bool((d <= 2.0 ** -7 * full["score1"].float().abs().cpu().clamp_min(0.5)).all()), d


def test_row_trimming_rule_is_per_clip_in_a_mixed_batch():
    """which kernels finish a clip's consumed rows must follow from the clip alone.  A clip with 11 consumed rows (score row +
    10 answer rows) next to a clip with 31 (21 more answer rows unmasked in its labels): the first finishes on the weight-streaming path
    and the second on the tile kernels with every other row, alone AND together - scores and level tokens of both are bit-identical
    alone vs in the batch, in either order."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=2, T=2, seed=33)
    labels = toks["labels"].clone()
    ans = (labels[1] != -100).nonzero().flatten()
    extra = torch.arange(int(ans[0]) - 21, int(ans[0]))                     # 21 more consumed rows in front of clip 1's answer
    labels[1, extra] = toks["input_ids"][1, extra]
    assert int((labels[0] != -100).sum()) + 1 <= 16 < int((labels[1] != -100).sum()) + 1
    n1 = toks["input_ids"].shape[1] - 1

    def run(order):
        idx = torch.tensor(order)
        fidx = (idx[:, None] * 2 + torch.arange(2)[None, :]).flatten()
        o = model(mos=None, pixel_values=pv[fidx], input_ids=toks["input_ids"][idx], attention_mask=toks["attention_mask"][idx],
                  image_flags=torch.ones(len(fidx), 1, dtype=torch.long), labels=labels[idx], motion_feature=motion[idx])
        torch.cuda.synchronize()
        return {b: (o["score1"][i:i + 1].clone(), o["logit"][i * n1:(i + 1) * n1].clone()) for i, b in enumerate(order)}

    alone = {**run([0]), **run([1])}
    for order in ([0, 1], [1, 0]):
        both = run(order)
        for b in (0, 1):
            assert torch.equal(both[b][0], alone[b][0]), (order, b, both[b][0], alone[b][0])
            assert torch.equal(both[b][1], alone[b][1]), (order, b)
    assert int((alone[1][1] >= 0).sum()) == 31 and int((alone[0][1] >= 0).sum()) == 10


def test_shared_prefix_scoring_matches_separate_passes():
    """Four prompts behind one video prefix: the prefix runs once into the KV cache, every prompt continues
    its own question / answer tokens (aigv_llm_extend).  Each result must be that of a separate full pass: identical level
    tokens (a rounding tie may flip), score within one bf16 ulp; and prompt 0 still matches the oracle."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    model, sd, toks, pv, motion, ref, out0 = run_case(cfg, B=2, T=2, seed=31)
    prompts = synth.perspective_prompts(toks, 4, seed=31)
    flags = torch.ones(pv.shape[0], 1, dtype=torch.long)
    outs = model.forward_shared_prefix([(p["input_ids"], p["attention_mask"], p["labels"]) for p in prompts], pixel_values=pv,
                                       image_flags=flags, motion_feature=motion)
    torch.cuda.synchronize()
    assert len(outs) == 4
    check_levels(outs[0], ref)
    score_ok(outs[0]["score1"], ref["score1"])
    flips = 0
    for p, got in zip(prompts, outs):
        sep = model(mos=None, pixel_values=pv, input_ids=p["input_ids"], attention_mask=p["attention_mask"], image_flags=flags,
                    labels=p["labels"], motion_feature=motion)
        assert torch.equal(got["label"], sep["label"]) and got["logit"].shape == sep["logit"].shape
        want = (p["labels"][:, 1:] != -100).reshape(-1)
        flips += int((got["logit"].cpu()[want] != sep["logit"].cpu()[want]).sum())
        assert bool((got["logit"].cpu()[~want] == -1).all())
        d = (got["score1"].float() - sep["score1"].float()).abs().cpu()
        assert bool((d <= 2.0 ** -7 * sep["score1"].float().abs().cpu().clamp_min(0.5)).all()), (d, got["score1"], sep["score1"])
    assert flips <= 1, flips


def test_shared_prefix_with_ragged_clips_and_prompts():
    """Clips with different prefix lengths (clip 1 drops 7 leading text tokens and is right-padded) under prompts of different
    lengths: per-sequence key offsets, positions and cache slots of the continuation pass.  Against separate passes."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    seed = 41
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    model = make_model(cfg, sd)
    base = synth.canonical_tokens(cfg, 2, 2, seed=seed)
    model.img_context_token_id = base["img_context_token_id"]
    prompts = synth.perspective_prompts(base, 3, seed=seed, question_lens=(16, 9, 23))
    drop = 7
    ragged = []
    for p in prompts:
        ids, lab = p["input_ids"], p["labels"]
        n = ids.shape[1]
        ids1 = torch.cat([ids[1, drop:], torch.zeros(drop, dtype=torch.long)])
        lab1 = torch.cat([lab[1, drop:], torch.full((drop,), -100)])
        am = torch.ones(2, n, dtype=torch.bool)
        am[1, n - drop:] = False
        ragged.append((torch.stack([ids[0], ids1]), am, torch.stack([lab[0], lab1])))
    pv = synth.synthetic_frames(4, 224, seed=seed)
    motion = synth.synthetic_motion(2, cfg.motion_dim, seed=seed)
    flags = torch.ones(4, 1, dtype=torch.long)
    outs = model.forward_shared_prefix(ragged, pixel_values=pv, image_flags=flags, motion_feature=motion)
    flips = 0
    for (ids, am, lab), got in zip(ragged, outs):
        sep = model(mos=None, pixel_values=pv, input_ids=ids, attention_mask=am, image_flags=flags, labels=lab, motion_feature=motion)
        want = (lab[:, 1:] != -100).reshape(-1)
        assert int(want.sum()) == 20
        flips += int((got["logit"].cpu()[want] != sep["logit"].cpu()[want]).sum())
        assert bool((got["logit"].cpu()[~want] == -1).all())
        d = (got["score1"].float() - sep["score1"].float()).abs().cpu()
        assert bool((d <= 2.0 ** -7 * sep["score1"].float().abs().cpu().clamp_min(0.5)).all()), (d, got["score1"], sep["score1"])
    assert flips <= 1, flips
    # prompts that diverge inside the video tokens cannot share a prefix
    bad = ragged[1][0].clone()
    bad[0, 41] = 5 if int(bad[0, 41]) != 5 else 6          # a text token in front of the first frame's <img>
    with pytest.raises(ValueError, match="diverge"):
        model.forward_shared_prefix([ragged[0], (bad, ragged[1][1], ragged[1][2])], pixel_values=pv, image_flags=flags, motion_feature=motion)


def test_shared_prefix_with_a_long_continuation():
    """A 210-token question: the continuation spans two 128-row query blocks of the attention kernel, each with the cached
    prefix in front (key offsets + causal masking across the block boundary)."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    seed = 43
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    model = make_model(cfg, sd)
    base = synth.canonical_tokens(cfg, 2, 2, seed=seed)
    model.img_context_token_id = base["img_context_token_id"]
    prompts = synth.perspective_prompts(base, 2, seed=seed, question_lens=(16, 210))
    pv = synth.synthetic_frames(4, 224, seed=seed)
    motion = synth.synthetic_motion(2, cfg.motion_dim, seed=seed)
    flags = torch.ones(4, 1, dtype=torch.long)
    outs = model.forward_shared_prefix([(p["input_ids"], p["attention_mask"], p["labels"]) for p in prompts], pixel_values=pv,
                                       image_flags=flags, motion_feature=motion)
    flips = 0
    for p, got in zip(prompts, outs):
        sep = model(mos=None, pixel_values=pv, input_ids=p["input_ids"], attention_mask=p["attention_mask"], image_flags=flags,
                    labels=p["labels"], motion_feature=motion)
        want = (p["labels"][:, 1:] != -100).reshape(-1)
        flips += int((got["logit"].cpu()[want] != sep["logit"].cpu()[want]).sum())
        d = (got["score1"].float() - sep["score1"].float()).abs().cpu()
        assert bool((d <= 2.0 ** -7 * sep["score1"].float().abs().cpu().clamp_min(0.5)).all()), (d, got["score1"], sep["score1"])
    assert flips <= 1, flips
    # the long prompt also agrees with the oracle run on its own
    ref = O.forward_eval(sd, cfg, pv, prompts[1]["input_ids"], prompts[1]["attention_mask"], flags, prompts[1]["labels"], motion,
                         prompts[1]["img_context_token_id"], mos=None, stage=2, return_intermediates=True)
    check_levels(outs[1], ref)
    score_ok(outs[1]["score1"], ref["score1"])


def test_extend_continues_a_prefill_exactly_like_a_longer_prefill():
    """aigv_llm_extend with commit: prefill(prompt[:-k]) + extend(prompt[-k:]) yields the next token of prefill(prompt), and a
    decode step after the committed extension agrees with the decode step after the full prefill."""
    from aigv_assessor_amd import native
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    model, sd, toks, pv, motion, ref, out0 = run_case(cfg, B=2, T=2, seed=33)
    lib = native.load()
    flags = torch.ones(pv.shape[0], 1, dtype=torch.long)
    plan = model._plan(toks["input_ids"], toks["attention_mask"], toks["labels"], flags, pv.shape[0])
    vit_embeds, mot = model._visual_inputs(pv, None, motion, plan)
    B, cu, k = 2, plan["cu"], 7
    last_rows = [cu[b + 1] - 1 for b in range(B)]
    cap = max(plan["lens"]) + 8
    _, nxt_full = model._prefill(plan["ids_packed"], plan["slot"], cu, vit_embeds, plan["n_vis"], mot, None, last_rows, keep_kv=True, kv_cap=cap)
    ctx = model._ctx        # (the context is re-created when the KV capacity grows: take the handle after the prefill)
    step_full = torch.empty_like(nxt_full)
    native.check(lib.aigv_decode_step(ctx, nxt_full.contiguous().data_ptr(), step_full.data_ptr(), None), ctx)
    torch.cuda.synchronize()
    # the same prompt as prefix + k-token extension
    ids_p = torch.cat([plan["ids_packed"][cu[b]:cu[b + 1] - k] for b in range(B)])
    slot_p = torch.cat([plan["slot"][cu[b]:cu[b + 1] - k] for b in range(B)])
    cu_p = [0, plan["lens"][0] - k, plan["lens"][0] + plan["lens"][1] - 2 * k]
    model._prefill(ids_p, slot_p, cu_p, vit_embeds, plan["n_vis"], mot, None, [], keep_kv=True, kv_cap=cap)
    tail = torch.cat([plan["ids_packed"][cu[b + 1] - k:cu[b + 1]] for b in range(B)]).to(torch.long).cuda()
    nxt_ext = torch.empty(B, dtype=torch.long, device="cuda")
    native.check(lib.aigv_llm_extend(ctx, tail.data_ptr(), native.i32_array([0, k, 2 * k]), B, None, None,
                                     native.i32_array([k - 1, 2 * k - 1]), B, nxt_ext.data_ptr(), 1, None), ctx)
    step_ext = torch.empty_like(nxt_ext)
    native.check(lib.aigv_decode_step(ctx, nxt_ext.contiguous().data_ptr(), step_ext.data_ptr(), None), ctx)
    torch.cuda.synchronize()
    assert torch.equal(nxt_ext.cpu(), nxt_full.cpu()), (nxt_ext, nxt_full)
    assert torch.equal(step_ext.cpu(), step_full.cpu()), (step_ext, step_full)
    # without a cache the call is refused
    model._prefill(plan["ids_packed"], plan["slot"], cu, vit_embeds, plan["n_vis"], mot, None, last_rows)
    assert lib.aigv_llm_extend(ctx, tail.data_ptr(), native.i32_array([0, k, 2 * k]), B, None, None, native.i32_array([0, k]), B,
                               nxt_ext.data_ptr(), 0, None) != 0


def test_stage1_and_ragged_padded_batch():
    cfg = pkg.tiny(image_size=224, vit_layers=1)
    seed = 9
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    model = make_model(cfg, sd, stage=1)
    # two clips with different answer lengths, right-padded to a common N like the training collator would
    t0 = synth.canonical_tokens(cfg, 1, 4, seed=1, answer_len=9)
    t1 = synth.canonical_tokens(cfg, 1, 4, seed=2, answer_len=5)
    n = t0["input_ids"].shape[1]
    pad = n - t1["input_ids"].shape[1]
    ids = torch.cat([t0["input_ids"], torch.cat([t1["input_ids"], torch.zeros(1, pad, dtype=torch.long)], 1)])
    labels = torch.cat([t0["labels"], torch.cat([t1["labels"], torch.full((1, pad), -100)], 1)])
    am = torch.cat([torch.ones(1, n, dtype=torch.bool), torch.cat([torch.ones(1, n - pad, dtype=torch.bool), torch.zeros(1, pad, dtype=torch.bool)], 1)])
    pv = synth.synthetic_frames(8, 224, seed=seed)
    motion = synth.synthetic_motion(2, cfg.motion_dim, seed=seed)
    model.img_context_token_id = t0["img_context_token_id"]
    out = model(pixel_values=pv, input_ids=ids, attention_mask=am, image_flags=torch.ones(8, 1, dtype=torch.long),
                labels=labels, motion_feature=motion)
    assert "score1" not in out
    for b, t in enumerate((t0, t1)):      # oracle per clip, un-padded
        ref = O.forward_eval(sd, cfg, pv[4 * b:4 * b + 4], t["input_ids"], t["attention_mask"], torch.ones(4, 1, dtype=torch.long),
                             t["labels"], motion[b:b + 1], t["img_context_token_id"], stage=1)
        nb = t["input_ids"].shape[1]
        got = out["logit"].view(2, n - 1)[b, : nb - 1].cpu()
        want = ref["label"] != -100
        assert torch.equal(got[want], ref["logit"][want])


def test_rmsnorm_qknorm_vit_flavour():
    cfg = pkg.tiny(image_size=224, norm_type="rms_norm", qk_norm=True, qkv_bias=False, llm_layers=1)
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=1, T=4, seed=10)
    rel_close(model.vit_tokens(pv), O.shuffled_tokens(O.vit_forward(sd, cfg, pv)), 0.01, 0.3)
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])


@pytest.mark.parametrize("max_pos", [256, 128])
def test_dynamic_ntk_rope_is_keyed_on_the_sequence_length_not_the_packed_batch(max_pos):
    """Real checkpoints ship rope_scaling = {dynamic, 2.0} (internvl_chat_eval2/config.json:82-85).  The reference rescales the
    rotary base when ONE sequence is longer than max_position_embeddings (modeling_internlm2.py:218-243).  Two clips of 215
    tokens pack into 430 rows: with max_pos 256 nothing may be rescaled (the packed count is not a sequence length); with
    max_pos 128 every clip is, by the base of its own length.  Both against the oracle, which follows the reference."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    cfg.llm_config.rope_scaling = {"type": "dynamic", "factor": 2.0}
    cfg.llm_config.max_position_embeddings = max_pos
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=2, T=2, seed=51)
    assert toks["input_ids"].shape[1] == 215
    assert model._rope_ntk == (215 if max_pos == 128 else 0)
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])
    # the two settings really differ: the same inputs with plain tables give other hidden states
    cfg.llm_config.rope_scaling = None
    plain = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], torch.ones(4, 1, dtype=torch.long), toks["labels"],
                           motion, toks["img_context_token_id"], stage=2, return_intermediates=True)
    same = torch.equal(plain["score1"], ref["score1"]) and torch.equal(plain["logit"], ref["logit"])
    assert same == (max_pos == 256)


@pytest.mark.parametrize("case", ["cross", "beyond"])
def test_decode_past_max_positions_with_dynamic_ntk(golden_dir, case):
    """Decoding past max_position_embeddings with rope_scaling = dynamic (real checkpoints ship it): at every such step the reference
    rebuilds its rotary tables with the base of the current kv_seq_len and rotates only the NEW token with them; cached keys keep the base
    they were written with (modeling_internlm2.py:187-194,227-243).  Tokens recorded from the reference's own InternLM2 cache path
    (tests/golden/ntk_decode.pt; the oracle reproduces them exactly: tests/test_oracle_golden.py) are fed back step by step (teacher
    forcing through the logits-processor hook of the decode loop), so every step is checked on the reference's own history: its logits
    against the oracle's, its choice against the reference's wherever the reference's top-2 gap is not a near-tie.  The same decode
    with plain tables is several times farther from the oracle than the HIP path is (the check has power)."""
    g = torch.load(os.path.join(golden_dir, "ntk_decode.pt"), weights_only=True)
    c, L = g["cases"][case], g["llm_config"]
    cfg = pkg.tiny(llm_hidden=L["hidden_size"], llm_heads=L["num_attention_heads"], llm_kv_heads=L["num_key_value_heads"],       # (the fixture's
                   llm_layers=L["num_hidden_layers"], llm_inter=L["intermediate_size"], vocab=L["vocab_size"])                    # configuration: its weights)
    cfg.llm_config.max_position_embeddings = L["max_position_embeddings"]
    cfg.llm_config.rope_scaling = dict(L["rope_scaling"])
    sd = synth.make_state_dict(cfg, seed=g["seed"], rich=True)
    b, n, new = c["b"], c["prompt"], c["new"]
    ref_tok = c["tokens"]

    def oracle_logits(scaling):
        """per-step logits of the oracle's cache path on the reference's token history"""
        cfg.llm_config.rope_scaling = scaling
        mask = torch.ones(b, n, dtype=torch.long)
        hidden, past, _ = O.llm_forward(sd, cfg, torch.nn.functional.embedding(c["ids"], O.embed_weight(sd)), mask.bool(), mask.cumsum(-1) - 1)
        rows = []
        for t in range(new):
            rows.append(O.lm_logits(sd, hidden[:, -1:, :])[:, -1, :].float())
            mask = torch.cat([mask, torch.ones(b, 1, dtype=torch.long)], 1)
            hidden, past, _ = O.llm_forward(sd, cfg, torch.nn.functional.embedding(ref_tok[:, t:t + 1], O.embed_weight(sd)), mask.bool(),
                                            (mask.cumsum(-1) - 1)[:, -1:], past)
        cfg.llm_config.rope_scaling = dict(L["rope_scaling"])
        return torch.stack(rows, 1)                              # [b, new, V]

    want = oracle_logits(dict(L["rope_scaling"]))
    solid = c["top2_gap"] >= 0.03                                 # (on the recording host the oracle reproduces every token: test_oracle_golden.py;
    assert torch.equal(want.argmax(-1)[solid], ref_tok[solid])    #  another host's bf16 matmul may turn a one-ulp tie the other way)
    plain = oracle_logits(None)
    model = make_model(cfg, sd)
    seen = []

    def force(hist, logits):                                      # record the step's logits, then continue on the reference's token
        t = hist.shape[1]
        seen.append(logits.float().cpu())
        out = torch.full_like(logits, float("-inf"))
        out[torch.arange(b), ref_tok[:, t].to(logits.device)] = 0
        return out

    cu = [i * n for i in range(b + 1)]
    got = model._greedy(c["ids"].reshape(-1), torch.full((b * n,), -1, dtype=torch.int32), cu, None, 0, new, [], 0, processors=[force])
    assert torch.equal(got.cpu(), ref_tok)
    have = torch.stack(seen, 1)
    assert model._rope_ntk == n + new - 1                         # the tables were rebuilt for the last decoded position's kv length
    scale = want.abs().max().item()
    err = (have - want).abs().max().item()
    past = max(0, L["max_position_embeddings"] - n)               # first decode step whose kv length is past the limit
    err_mean = (have[:, past:] - want[:, past:]).abs().mean().item()
    power = (plain[:, past:] - want[:, past:]).abs().mean().item()
    print(f"{case}: |hip - oracle| logits max {err:.4f} mean {err_mean:.5f} past the limit (logit scale {scale:.2f}); the plain-table decode differs by mean {power:.5f}")
    assert err <= 0.02 * scale and power >= 5 * err_mean
    clear = c["top2_gap"] >= 8 * err                              # the reference's choice is no near-tie at the path's own accuracy
    assert torch.equal(have.argmax(-1)[clear], ref_tok[clear]) and int(clear.sum()) >= new // 3


def test_greedy_generate_matches_oracle_cache_path():
    cfg = pkg.tiny(image_size=224)
    seed = 11
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, 2, 4, seed=seed)
    n_prompt = int((toks["labels"][0] == -100).sum())
    ids = toks["input_ids"][:, :n_prompt].clone()
    ctx = toks["img_context_token_id"]
    for b in range(2):   # generate() prompts carry no motion slot: turn the lone trailing <IMG_CONTEXT> into text
        ids[b, (ids[b] == ctx).nonzero()[-1]] = 7
    pv = synth.synthetic_frames(8, 224, seed=seed)
    vit = O.extract_feature(sd, cfg, pv)
    emb = O.scatter_embeds(sd, ids, ctx, vit, None)
    want = O.greedy_generate(sd, cfg, emb, torch.ones_like(ids), max_new_tokens=6)
    model = make_model(cfg, sd)
    model.img_context_token_id = ctx
    got = model.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=6, do_sample=False)
    assert torch.equal(got.cpu(), want), (got.cpu().tolist(), want.tolist())
    # generate2 from the oracle's embeddings must give the same continuation
    got2 = model.generate2(input_embeds=emb, attention_mask=torch.ones_like(ids), max_new_tokens=6)
    assert torch.equal(got2.cpu(), want)


@pytest.mark.parametrize("heads,kv", [(6, 2), (5, 1), (7, 1)])
def test_greedy_generate_with_odd_gqa_groups(heads, kv):
    """Round 6 (tests/manual/fuzz_generate.py): the decode attention was instantiated for 1 / 2 / 4 / 6 / 8 query heads per KV head only - a
    configuration with 3 (or 5, 7) scored fine (the prefill kernel takes any group) and failed with 'invalid argument' at its first decode step.
    Now 1..8 are built (more are refused when the context is created).  Teacher-forced against the oracle's cache path: every generated token is
    the oracle's argmax at its step or within 2 bf16 ulps of it."""
    cfg = pkg.tiny(image_size=224, llm_hidden=128 * heads, llm_heads=heads, llm_kv_heads=kv, llm_layers=2, llm_inter=512)
    seed, B, T, n_new = 40 + heads, 3, 2, 5
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    n_prompt = int((toks["labels"][0] == -100).sum())
    ids = toks["input_ids"][:, :n_prompt].clone()
    ctx = toks["img_context_token_id"]
    for b in range(B):
        ids[b, (ids[b] == ctx).nonzero()[-1]] = 7
    pv = synth.synthetic_frames(B * T, 224, seed=seed)
    model = make_model(cfg, sd)
    model.img_context_token_id = ctx
    got = model.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=n_new, do_sample=False).cpu()
    emb = O.scatter_embeds(sd, ids, ctx, O.extract_feature(sd, cfg, pv), None)
    exact, gaps = _teacher_forced_gaps(sd, cfg, emb, torch.ones_like(ids), got)
    print(f"GQA group {heads // kv}: {exact}/{B * n_new} tokens are the oracle's argmax, gaps {gaps}")
    assert got.shape == (B, n_new) and all(x <= 2 for x in gaps) and exact >= B * n_new - 3, gaps
    if heads == 7:          # more than 8 query heads per KV head: refused when the context is created, with a message that says why
        from aigv_assessor_amd import native
        big = pkg.tiny(image_size=224, llm_hidden=128 * 9, llm_heads=9, llm_kv_heads=1, llm_layers=1, llm_inter=256)
        m9 = make_model(big, synth.make_state_dict(big, seed=1, rich=True))
        m9.img_context_token_id = ctx
        t9 = synth.canonical_tokens(big, 1, 1, seed=1)
        with pytest.raises(native.NativeError, match="GQA groups of more than 8"):
            m9(mos=None, pixel_values=synth.synthetic_frames(1, 224, seed=1), input_ids=t9["input_ids"], attention_mask=t9["attention_mask"],
               image_flags=torch.ones(1, 1, dtype=torch.long), labels=t9["labels"], motion_feature=synth.synthetic_motion(1, big.motion_dim, seed=1))


@pytest.mark.parametrize("B,T", [(6, 1), (20, 1)])
def test_greedy_generate_with_many_sequences(B, T):
    """Decode steps with more sequences than the small-batch forms take (<= 4: fused-norm / 4-row forms, <= 8: 8-row forms, <= 16: one
    row tile, above: several row tiles per GEMV) - positions, KV slots and the split-KV attention per sequence - against the oracle."""
    cfg = pkg.tiny(image_size=224)
    seed = 12
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    n_prompt = int((toks["labels"][0] == -100).sum())
    ids = toks["input_ids"][:, :n_prompt].clone()
    ctx = toks["img_context_token_id"]
    for b in range(B):
        ids[b, (ids[b] == ctx).nonzero()[-1]] = 7
    pv = synth.synthetic_frames(B * T, 224, seed=seed)
    emb = O.scatter_embeds(sd, ids, ctx, O.extract_feature(sd, cfg, pv), None)
    want = O.greedy_generate(sd, cfg, emb, torch.ones_like(ids), max_new_tokens=5)
    model = make_model(cfg, sd)
    model.img_context_token_id = ctx
    got = model.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=5, do_sample=False).cpu()
    same = (got == want).all(dim=1)
    print(f"B={B}: {int(same.sum())}/{B} sequences token-identical to the oracle")
    # a sequence may leave the oracle's path at a near-tie of the logits (bf16 noise of different summation orders); most must not
    assert int(same.sum()) >= B - max(1, B // 5), (got.tolist(), want.tolist())
    assert torch.equal(got[:, 0], want[:, 0]) or int((got[:, 0] != want[:, 0]).sum()) <= 1


def test_against_reference_golden_vectors(golden_dir):
    """The fixtures were produced by the imported REFERENCE (tests/golden/make_golden.py), not by the oracle."""
    e2e = torch.load(os.path.join(golden_dir, "e2e.pt"), weights_only=True)
    anchors = torch.load(os.path.join(golden_dir, "e2e_anchors.pt"), weights_only=True)
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=e2e["vision_config"], llm_config=e2e["llm_config"],
                                                force_image_size=448, select_layer=-1))
    for tag in ("bf16_b1", "bf16_b2"):
        g = e2e[tag]
        B, T, seed = g["B"], g["T"], g["seed"]
        sd = synth.make_state_dict(cfg, seed=seed, rich=True)
        toks = synth.canonical_tokens(cfg, B, T, seed=seed)
        model = make_model(cfg, sd)
        model.img_context_token_id = toks["img_context_token_id"]
        out = model(mos=torch.full((B,), 0.5, dtype=BF), pixel_values=synth.synthetic_frames(B * T, 448, seed=seed),
                    input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                    image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=toks["labels"],
                    motion_feature=synth.synthetic_motion(B, 2304, seed=seed))
        want = g["label"] != -100
        # the tie rule and the fp32 anchor read the reference's own recorded logits / fp32 pass (e2e_anchors.pt; a live oracle pass until round 5)
        a = anchors[f"internlm2/{seed}"]
        assert assert_levels_recorded(out["logit"].cpu()[want], g["logit"][want], a["top_ids"], a["top_values"]) <= 2
        score_ok(out["score1"], g["score1"], ulps=4)       # 4096-wide LLM: see score_near_fp32
        score_near_recorded_fp32(out["score1"], g["score1"], a["score1_fp32_of_bf16_weights"])
        del model
        torch.cuda.empty_cache()


def test_llama_family_against_the_reference_golden_vectors(golden_dir):
    """The reference's second LLM family (modeling_internvl_chat.py:228-229: transformers' LlamaForCausalLM): a checkpoint with HF Llama
    tensor names, loaded through load_state_dict (re-packed on the host: weights.llama_to_internlm2), against the outputs the REFERENCE
    recorded with that LLM class (tests/golden/make_golden_llama.py) - same bars as the InternLM2 fixture of the same widths."""
    from aigv_assessor_amd import weights
    g = torch.load(os.path.join(golden_dir, "e2e_llama.pt"), weights_only=True)
    anchors = torch.load(os.path.join(golden_dir, "e2e_anchors.pt"), weights_only=True)
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=g["vision_config"], llm_config=g["llm_config"], force_image_size=448, select_layer=-1))
    for tag in ("bf16_b1", "bf16_b2"):
        c = g[tag]
        B, T, seed = c["B"], c["T"], c["seed"]
        sd = weights.internlm2_to_llama(synth.make_state_dict(cfg, seed=seed, rich=True), cfg.llm_config)
        toks = synth.canonical_tokens(cfg, B, T, seed=seed)
        pv, motion, flags = synth.synthetic_frames(B * T, 448, seed=seed), synth.synthetic_motion(B, 2304, seed=seed), torch.ones(B * T, 1, dtype=torch.long)
        model = make_model(cfg, sd)
        assert model.llm_arch_name == "LlamaForCausalLM"
        model.img_context_token_id = toks["img_context_token_id"]
        out = model(mos=torch.full((B,), 0.5, dtype=BF), pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                    image_flags=flags, labels=toks["labels"], motion_feature=motion)
        want = c["label"] != -100
        a = anchors[f"llama/{seed}"]        # the reference's recorded answer-row logits and fp32 pass (a live oracle pass until round 5)
        assert assert_levels_recorded(out["logit"].cpu()[want], c["logit"][want], a["top_ids"], a["top_values"]) <= 2
        score_ok(out["score1"], c["score1"], ulps=4)
        score_near_recorded_fp32(out["score1"], c["score1"], a["score1_fp32_of_bf16_weights"])
        if tag == "bf16_b1":   # greedy decode through the KV cache against the reference LLM's own cache path
            n_prompt = c["greedy_prompt_len"]
            ids = toks["input_ids"][:, :n_prompt]
            vit = O.extract_feature(sd, cfg, pv).reshape(-1, 4096)
            emb = O.scatter_embeds(sd, ids, toks["img_context_token_id"], torch.cat([vit, vit[:1]]), None)
            got = model.generate2(input_embeds=emb, attention_mask=torch.ones(1, n_prompt, dtype=torch.long), max_new_tokens=6)
            assert torch.equal(got.cpu(), c["greedy_tokens"]), (got.cpu().tolist(), c["greedy_tokens"].tolist())
        del model
        torch.cuda.empty_cache()


def test_llama_family_streamed_load_equals_the_dict_load():
    """load_state_dict_stream with HF Llama names (q / k / v arrive one by one and leave as one packed wqkv) == load_state_dict."""
    from aigv_assessor_amd import weights
    from aigv_assessor_amd.modeling import InternVLChatModel
    cfg = pkg.tiny(image_size=224)
    cfg.llm_config.architectures = ("LlamaForCausalLM",)
    packed = synth.make_state_dict(cfg, seed=3, rich=True)
    sd = weights.internlm2_to_llama(packed, cfg.llm_config)
    a = make_model(cfg, sd)
    b = InternVLChatModel(cfg)
    assert b.load_state_dict_stream(iter(sd.items())) == []
    b = b.eval().cuda()
    toks = synth.canonical_tokens(cfg, 2, 4, seed=3)
    kw = dict(pixel_values=synth.synthetic_frames(8, 224, seed=3), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
              image_flags=torch.ones(8, 1, dtype=torch.long), labels=toks["labels"], motion_feature=synth.synthetic_motion(2, cfg.motion_dim, seed=3))
    for m in (a, b):
        m.img_context_token_id = toks["img_context_token_id"]
    oa, ob = a(**kw), b(**kw)
    assert torch.equal(oa["score1"], ob["score1"]) and torch.equal(oa["logit"], ob["logit"])
    with pytest.raises(RuntimeError):      # Llama names into an InternLM2 configuration
        cfg2 = pkg.tiny(image_size=224)
        InternVLChatModel(cfg2).load_state_dict(sd)


def test_score_accuracy_matches_reference_bf16_path():
    """Over several seeds the HIP score must be as close to the fp32 oracle as the reference-faithful bf16 oracle is."""
    cfg = pkg.tiny(image_size=224, llm_layers=3, vit_layers=3)
    B, T = 1, 4
    e_hip, e_ref, same = [], [], 0
    for seed in range(31, 37):
        sd = synth.make_state_dict(cfg, seed=seed, rich=True)
        toks = synth.canonical_tokens(cfg, B, T, seed=seed)
        pv, mo = synth.synthetic_frames(B * T, 224, seed=seed), synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
        flags = torch.ones(B * T, 1, dtype=torch.long)
        args = (toks["input_ids"], toks["attention_mask"], flags, toks["labels"])
        f32 = O.forward_eval({k: v.float() for k, v in sd.items()}, cfg, pv.float(), *args, mo.float(), toks["img_context_token_id"])["score1"].item()
        b16 = O.forward_eval(sd, cfg, pv, *args, mo, toks["img_context_token_id"])["score1"].float().item()
        model = make_model(cfg, sd)
        model.img_context_token_id = toks["img_context_token_id"]
        hip = model(pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                    labels=toks["labels"], motion_feature=mo)["score1"].float().item()
        e_hip.append(abs(hip - f32)); e_ref.append(abs(b16 - f32)); same += hip == b16
        del model
    print(f"mean|hip-fp32| {sum(e_hip) / 6:.5f} mean|bf16ref-fp32| {sum(e_ref) / 6:.5f} identical-bf16 {same}/6")
    assert sum(e_hip) / 6 <= 1.5 * sum(e_ref) / 6 + 1e-3
    assert same >= 3


def test_errors_are_loud():
    from aigv_assessor_amd import native
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=1)
    sd = synth.make_state_dict(cfg, seed=1)
    model = make_model(cfg, sd)
    with pytest.raises(ValueError):
        model.extract_feature(torch.zeros(3, 224, 224))              # wrong rank (modeling_intern_vit.py:345)
    with pytest.raises(AssertionError):
        model.generate(pixel_values=None, input_ids=torch.zeros(1, 4, dtype=torch.long))   # img_context_token_id unset
    toks = synth.canonical_tokens(cfg, 1, 4, seed=1)
    model.img_context_token_id = toks["img_context_token_id"]
    with pytest.raises(ValueError):                                   # 3 frames for a 4-frame prompt
        model(pixel_values=synth.synthetic_frames(3, 224), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
              image_flags=torch.ones(3, 1, dtype=torch.long), labels=toks["labels"],
              motion_feature=synth.synthetic_motion(1, cfg.motion_dim))
    with pytest.raises(RuntimeError):                                 # SlowFast is an input
        model(pixel_values=synth.synthetic_frames(4, 224), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
              image_flags=torch.ones(4, 1, dtype=torch.long), labels=toks["labels"])


def test_26b_widths_and_16_frames_smoke():
    """BASELINE configs 3/4 plumbing at reduced depth: InternViT-6B widths (RMSNorm + QK-norm, 25 heads x 128) and
    InternLM2-20B widths (48 q / 8 kv heads), 16 frames per clip -> N = 4281 tokens; parity vs the oracle on levels/score."""
    cfg = pkg.internvl2_26b()
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 1
    cfg.llm_config.vocab_size = 2048
    cfg.vision_config.intermediate_size = 1280
    cfg.llm_config.intermediate_size = 2048
    cfg.score_dims = (256, 64, 16, 1)
    B, T, seed = 1, 16, 12
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    assert toks["input_ids"].shape[1] == 4281
    pv = synth.synthetic_frames(B * T, 448, seed=seed)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    ref = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                         toks["img_context_token_id"], stage=2, return_intermediates=True)
    model = make_model(cfg, sd)
    model.img_context_token_id = toks["img_context_token_id"]
    out = model(pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                labels=toks["labels"], motion_feature=motion)
    check_levels(out, ref)
    score_ok(out["score1"], ref["score1"])


def test_config4_26b_true_widths_stage1_16_frames():
    """BASELINE config 4 at its TRUE widths, reduced only in depth and vocabulary: InternViT-6B (hidden 3200, 25 heads x 128,
    intermediate 12800, RMSNorm + QK-norm) and InternLM2-20B (hidden 6144, 48 q / 8 kv heads, intermediate 16384), two layers each,
    stage-1 flavour (quality-level decode: no score head), one 16-frame 448-px clip -> N = 4281 tokens; against the oracle.
    (The reference model cannot be built at these widths: its score head and motion projector hard-code 4096)"""
    cfg = pkg.internvl2_26b()
    cfg.vision_config.num_hidden_layers = 2
    cfg.llm_config.num_hidden_layers = 2
    cfg.llm_config.vocab_size = 4096
    assert cfg.vision_config.intermediate_size == 12800 and cfg.llm_config.intermediate_size == 16384
    B, T, seed = 1, 16, 14
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    assert toks["input_ids"].shape[1] == 4281
    pv = synth.synthetic_frames(B * T, 448, seed=seed)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    ref = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                         toks["img_context_token_id"], stage=1, return_intermediates=True)
    model = make_model(cfg, sd, stage=1)
    model.img_context_token_id = toks["img_context_token_id"]
    out = model(pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                labels=toks["labels"], motion_feature=motion)
    assert "score1" not in out
    rel_close(model.vit_tokens(pv), O.shuffled_tokens(O.vit_forward(sd, cfg, pv)), 0.02, 0.5)
    check_levels(out, ref)


def test_config4_26b_full_depth_matches_the_oracle(golden_dir):
    """BASELINE config 4 at its true widths AND full depth (round 4): InternViT-6B x 45 layers + InternLM2-20B x 48
    layers, one 16-frame clip (N = 4281), stage-1 flavour - 26 G parameters, 51 GB of bf16 weights streamed from the seeded generator
    straight into the model (load_state_dict_stream).  ORACLE-ONLY: the fixture (tests/golden/make_golden_26b.py -> e2e_26b_full.pt) holds
    the oracle's bf16 and fp32 passes; the reference itself cannot be built at this width (its score head and motion projector hard-code
    a 4096-wide LLM, modeling_internvl_chat.py:44,244-249).  Bars as for the 8B fixtures: level tokens identical to the bf16 oracle or a
    near-tie of the oracle's own logits, agreement with the fp32 oracle no worse than the bf16 oracle's minus 4 rows, and the final hidden
    state of the score row position hidden[:, -4] at most 1.25 x as far (relative L2) from the fp32 oracle as the bf16 oracle is."""
    from aigv_assessor_amd.modeling import InternVLChatModel
    path = os.path.join(golden_dir, "e2e_26b_full.pt")
    if not os.path.exists(path):
        pytest.skip("tests/golden/e2e_26b_full.pt not generated")
    g = torch.load(path, weights_only=True)
    cfg = pkg.internvl2_26b()
    assert (g["vit_layers"], g["llm_layers"]) == (cfg.vision_config.num_hidden_layers, cfg.llm_config.num_hidden_layers) == (45, 48)
    T, N = g["T"], g["n_tokens"]
    dev = torch.device("cuda", 0)
    model = InternVLChatModel(cfg, device=dev, stage=1, max_clips=1, max_frames=T, max_tokens=N)
    # the fixture's weight set: since round 6 synth's "hash" method - the same bits on CPU (where the oracle recorded the fixture) and on the device,
    # where the 26 G parameters are now produced in seconds (the serial CPU generator of rounds 4-5 took 111 s of this test's 115)
    method = g.get("w_method", "randn")
    missing = model.load_state_dict_stream(synth.make_state_dict_iter(cfg, seed=g["w_seed"], rich=True, method=method, device=dev if method == "hash" else "cpu"))
    assert not missing, missing[:5]
    model.eval()
    toks = synth.canonical_tokens(cfg, 1, T, seed=g["in_seed"])
    assert toks["input_ids"].shape[1] == N == 4281
    model.img_context_token_id = toks["img_context_token_id"]
    out = model(pixel_values=synth.synthetic_frames(T, 448, seed=g["in_seed"]).to(dev), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                image_flags=torch.ones(T, 1, dtype=torch.long), labels=toks["labels"], motion_feature=synth.synthetic_motion(1, cfg.motion_dim, seed=g["in_seed"]).to(dev))
    torch.cuda.synchronize()
    assert "score1" not in out
    rows, r16, r32 = g["answer_rows"], g["cases"]["bf16"], g["cases"]["fp32"]
    assert torch.equal(out["label"].cpu()[rows], g["label"])
    got = out["logit"].cpu()[rows]
    n_tie = _level_rows_ok(got, r16, "26B full depth")
    agree_hip, agree_ref = int((got == r32["logit"]).sum()), int((r16["logit"] == r32["logit"]).sum())
    # hidden[:, -4] is the 8th of the ten consumed (answer) rows: rows N-11 .. N-2 of the shifted sequence
    assert int(rows[7]) == N - 4
    hid = model.last_hidden_rows(len(rows)).float().cpu()[7:8]
    h32 = r32["hidden_m4"].float()
    e_hip = float((hid - h32).norm() / h32.norm())
    e_ref = float((r16["hidden_m4"].float() - h32).norm() / h32.norm())
    print(f"26B full depth (oracle-only): level tokens {len(rows) - n_tie}/{len(rows)} identical to the bf16 oracle ({n_tie} near-ties), agreement with the fp32 "
          f"oracle: hip {agree_hip}, bf16 oracle {agree_ref}; hidden[:, -4] rel L2 vs fp32 oracle: hip {e_hip:.4f}, bf16 oracle {e_ref:.4f}")
    assert n_tie <= len(rows) // 4 + 1
    assert agree_hip >= agree_ref - 4
    assert e_hip <= 1.25 * e_ref, (e_hip, e_ref)
    del model
    torch.cuda.empty_cache()


def test_config5_fp8_mode_at_8b_widths():
    """BASELINE config 5's arithmetic at the 8B WIDTHS (hidden 4096, 32 q / 8 kv heads, intermediate 14336; three decoder layers, one
    InternViT-300M layer, reduced vocabulary), two 8-frame clips: the e4m3 linears against oracle/fp8.py - the kernel form that
    runs at full size (256-tile rounds + split-K tails at M = 4352), not the tiny-dimension instantiation of the test above."""
    from oracle import fp8 as O8
    cfg = pkg.internvl2_8b()
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 3
    cfg.llm_config.vocab_size = 4096
    B, T, seed = 2, 8, 15
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=B, T=T, seed=seed)             # bf16 pass + bf16 oracle
    check_levels(out, ref)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    score_ok(out["score1"], ref["score1"], ulps=4)         # 4096-wide LLM: see score_near_fp32
    score_near_fp32(out["score1"], ref["score1"], cfg, sd, toks, pv, motion, flags)
    with O8.fp8_llm(cfg.llm_config.num_hidden_layers):
        ref8 = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                              toks["img_context_token_id"], mos=None, stage=2, return_intermediates=True)
    kw = dict(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
              labels=toks["labels"], motion_feature=motion)
    model.set_precision("fp8")
    model.set_attention_numerics("reference")   # oracle/fp8.py rounds the score matrix like the reference: compare like with like (an e4m3 code is a 6 % step)
    try:
        out8 = model(**kw)
        all8 = model(full_logits=True, **kw)["logit"].cpu()              # argmax of EVERY row (debug surface)
        torch.cuda.synchronize()
    finally:
        model.set_attention_numerics("fp32")
        model.set_precision("bf16")
    # An e4m3 code is a 6 % step: a one-ulp difference in a bf16 activation (fp32 summation order, HIP vs CPU) can flip a code
    # downstream, so two correct evaluations of this arithmetic differ by more than two bf16 evaluations do.  Bars at
    # these widths: the score within 8 bf16 ulps of the mode's own oracle; over ALL rows the HIP argmax agrees with the fp8 oracle
    # more often than the fp8 oracle agrees with the bf16 oracle (it implements THIS arithmetic, not merely something as accurate);
    # on the answer rows every disagreement is a near-tie of the oracle's own logits.
    d8 = (out8["score1"].float().cpu() - ref8["score1"].float()).abs().max().item()
    agree_own = float((all8 == ref8["logit"]).float().mean())
    agree_modes = float((ref8["logit"] == ref["logit"]).float().mean())
    want = ref8["label"] != -100
    got, exp = out8["logit"].cpu()[want], ref8["logit"][want]
    logits = ref8["logits"][..., :-1, :].reshape(-1, ref8["logits"].shape[-1])[want]
    gaps = [round(abs(logits[r, exp[r]].item() - logits[r, got[r]].item()) / _bf16_ulp(logits[r, exp[r]].item()), 1)
            for r in (got != exp).nonzero().flatten().tolist()]
    drift = (out8["score1"].float().cpu() - ref["score1"].float()).abs().max().item()
    print(f"fp8 at 8B widths, 3 layers: |score - fp8 oracle| {d8:.4g}; drift vs bf16 {drift:.4g}; all-row argmax agreement hip fp8 vs fp8 oracle "
          f"{agree_own:.3f}, fp8 oracle vs bf16 oracle {agree_modes:.3f}; answer-row disagreements (oracle gap in bf16 ulps): {gaps}")
    assert d8 <= 8 * 2.0 ** -8
    assert agree_own >= agree_modes + 0.02
    assert len(gaps) <= max(1, int(want.sum()) // 4) and all(x <= 16 for x in gaps), gaps
    assert drift <= 0.08


def _teacher_forced_gaps(sd, cfg, emb, mask, tokens):
    """The oracle's KV-cache path fed the GIVEN tokens: per step, (is the token the oracle's argmax, the oracle's logit gap between its
    argmax and the token in bf16 ulps)."""
    B = tokens.shape[0]
    hidden, past, _ = O.llm_forward(sd, cfg, emb, mask.bool(), (mask.cumsum(-1) - 1))
    m = mask.clone()
    gaps, exact = [], 0
    for t in range(tokens.shape[1]):
        for b in range(B):
            logits = O.lm_logits(sd, hidden[b:b + 1, -1:, :])[0, -1].float()
            top, tok = int(logits.argmax()), int(tokens[b, t])
            exact += top == tok or float(logits[top]) == float(logits[tok])      # (a token whose bf16 logit EQUALS the maximum is an argmax too: argmax picks the first)
            gaps.append(round(abs(logits[top].item() - logits[tok].item()) / _bf16_ulp(logits[top].item()), 1))
        m = torch.cat([m, torch.ones((B, 1), dtype=m.dtype)], dim=1)
        emb_t = torch.nn.functional.embedding(tokens[:, t:t + 1], sd["language_model.model.tok_embeddings.weight"])
        hidden, past, _ = O.llm_forward(sd, cfg, emb_t, m.bool(), (m.cumsum(-1) - 1)[:, -1:], past)
    return exact, gaps


@pytest.mark.parametrize("B", [1, 3])
def test_decode_at_8b_widths_bf16_and_fp8_modes(B):
    """generate() at the 8B WIDTHS (hidden 4096, intermediate 14336; three decoder layers), where the decode step runs its
    width-specific forms - the GEMVs that apply the RMSNorm themselves, the sub-slab forms (csrc/head.hip), and in fp8 mode the
    e4m3 GEMVs that also quantise the token rows (csrc/head8.hip); the tiny configurations of the other decode tests take the generic
    forms.  Checked against the oracle's KV-cache path teacher-forced on the HIP tokens - oracle.py for bf16, oracle/fp8.py's
    arithmetic for fp8: every generated token is the oracle's argmax at that step or a near-tie of the oracle's own logits
    (random weights give near-uniform vocabulary logits; in fp8 an e4m3 code can flip on a one-ulp activation difference)."""
    from oracle import fp8 as O8
    cfg = pkg.internvl2_8b()
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 3
    cfg.llm_config.vocab_size = 4096
    # (2 frames per clip - a 599-token prompt: the decode-step forms under test depend on the widths and on B, not on the prompt length, and
    # the CPU oracle's prefill of the prompt is what this test's wall time was: 88 s of the driver's 900 s at 8 frames, round 5)
    T, seed, n_new = 2, 16, 6 if B == 1 else 4
    sd = synth.make_state_dict(cfg, seed=seed, rich=True)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    n_prompt = int((toks["labels"][0] == -100).sum())
    ids = toks["input_ids"][:, :n_prompt].clone()
    ctx = toks["img_context_token_id"]
    for b in range(B):
        ids[b, (ids[b] == ctx).nonzero()[-1]] = 7      # generate() prompts carry no motion slot
    pv = synth.synthetic_frames(B * T, 448, seed=seed)
    model = make_model(cfg, sd)
    model.img_context_token_id = ctx
    mask = torch.ones_like(ids)
    got_bf16 = model.generate(pixel_values=pv, input_ids=ids, attention_mask=mask, max_new_tokens=n_new, do_sample=False).cpu()
    model.set_precision("fp8")
    model.set_attention_numerics("reference")   # (the fp8 oracle's prefill rounds the score matrix like the reference)
    try:
        got_fp8 = model.generate(pixel_values=pv, input_ids=ids, attention_mask=mask, max_new_tokens=n_new, do_sample=False).cpu()
    finally:
        model.set_attention_numerics("fp32")
        model.set_precision("bf16")
    assert got_bf16.shape == (B, n_new) and got_fp8.shape == (B, n_new)
    emb = O.scatter_embeds(sd, ids, ctx, O.extract_feature(sd, cfg, pv), None)
    exact16, gaps16 = _teacher_forced_gaps(sd, cfg, emb, mask, got_bf16)
    with O8.fp8_llm(cfg.llm_config.num_hidden_layers):
        exact8, gaps8 = _teacher_forced_gaps(sd, cfg, emb, mask, got_fp8)
    print(f"decode at 8B widths: bf16 tokens {got_bf16.tolist()} oracle gaps (bf16 ulps) {gaps16} exact {exact16}/{B * n_new}; "
          f"fp8 tokens {got_fp8.tolist()} gaps {gaps8} exact {exact8}/{B * n_new}")
    assert all(x <= 2 for x in gaps16) and exact16 >= B * n_new - 2 * B, gaps16
    # fp8 mode: e4m3 noise (3.75 % per linear) puts the oracle's own candidates up to ~10 bf16 ulps apart where the HIP token differs (8.0 and 9.0 seen)
    assert all(x <= 12 for x in gaps8) and exact8 >= B * n_new // 2, gaps8


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json's headline configuration at FULL size (InternViT-300M x 24 layers + InternLM2.5-7B x 32 layers, 8 frames x
# 448 px, N = 2177), against outputs of the imported REFERENCE recorded by tests/golden/make_golden_8b.py.
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_8b(golden_dir):
    """The full-size model with the golden's seeded weights.  The weights come from the CPU generator (the values the
    reference run used; ~2 min for 8.1 G parameters) and are shared by the full-size tests of this module."""
    from aigv_assessor_amd.modeling import InternVLChatModel
    g = torch.load(os.path.join(golden_dir, "e2e_8b_full.pt"), weights_only=True)
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=g["vision_config"], llm_config=g["llm_config"],
                                                force_image_size=448, select_layer=-1))
    assert cfg.llm_config.num_hidden_layers == 32 and cfg.vision_config.num_hidden_layers == 24 and cfg.llm_config.rope_scaling
    dev = torch.device("cuda", 0)
    n_tok = synth.canonical_len(cfg, 8)
    model = InternVLChatModel(cfg, device=dev, max_clips=4, max_frames=32, max_tokens=4 * n_tok)
    sd = synth.make_state_dict(cfg, seed=g["w_seed"], rich=True)
    for k, v in g.get("overrides", {}).items():
        sd[k] = torch.full_like(sd[k], v)
    model.load_state_dict(sd)
    del sd
    model.eval()
    yield model, cfg, g
    del model
    torch.cuda.empty_cache()


def _golden_inputs(cfg, seed, dev):
    toks = synth.canonical_tokens(cfg, 1, 8, seed=seed)
    return toks, synth.synthetic_frames(8, 448, seed=seed).to(dev), synth.synthetic_motion(1, cfg.motion_dim, seed=seed).to(dev)


LEVEL_TIE_ULPS = 4.0   # (the hand-set near-tie bar of rounds 2-4; the full-size checks now read theirs from the fixture: _level_tie_bar)
# Full depth: |hip - reference bf16| per clip, in bf16 ulps of the score.  What two CORRECT evaluations of the reference's own arithmetic
# differ by was measured on the reference itself (rounds 4 and 5): the imported reference re-scores the first 13 recorded clips under
# torch.set_num_threads(1 / 2 / 4) - same weights, same torch build, only oneDNN's blocking, i.e. the fp32 summation order, changes - and
# its bf16 scores move against the recorded 8-thread pass by 2.56 bf16 ulps on average, 5.0 at the 95th percentile, 8.0 at most
# (44 pairs: tests/golden/e2e_8b_r5.pt from make_golden_8b_r5.py + the five pairs of e2e_8b_r4_self.pt; 4 of 44 identical; 22 of 390 level
# tokens flip).  A score difference is chaotic at this depth (one flipped rounding early on moves everything behind it).  The HIP path is one
# more evaluation of the same arithmetic: over the 37 reference-pinned clips it reads 2.80 mean / 9.0 max (profiles/r5_parity_stats.txt).
# Bars, READ FROM THE FIXTURES (factor <= 1.3, no hand-set number):
#   per clip      |hip - ref bf16| <= REF_SELF_FACTOR x the reference's own MAXIMUM  (1.3 x 8.0 = 10.4 ulps)
#   pooled mean   over all 37 clips  <= REF_SELF_FACTOR x the reference's own MEAN     (1.3 x 2.56 = 3.3 ulps)
# north_star's literal 1e-3 (a quarter of a bf16 ulp) lies below the reference's own reproducibility and is reported, not asserted.
REF_SELF_FACTOR = 1.3


def _ulps(d: float, ref: float) -> float:
    return abs(float(d)) / _bf16_ulp(float(ref))


_REF_SELF = {}
_REF_SELF_LEVELS = {}


def _ref_self_level_gap(golden_dir):
    """Level tokens of the reference against ITSELF (the same 39 thread-count pairs): (rows, rows whose token flips, the largest gap - in bf16
    ulps of the 8-thread pass's own logits - between its winner and the token another thread count picks, flips to a token outside its
    recorded top four).  Measured: 390 rows, 22 flips, largest gap 5.0, one outside the top four.  The full-size level checks accept a HIP
    token that differs from the reference's when the reference's own logits put it within REF_SELF_FACTOR x that largest gap (6.5 ulps) -
    the bar the reference would need against itself, read from the fixture (before round 5: a hand-set 4)."""
    if golden_dir in _REF_SELF_LEVELS:
        return _REF_SELF_LEVELS[golden_dir]
    ld = lambda f: torch.load(os.path.join(golden_dir, f), weights_only=True)["cases"]
    base = {f"one/{k.split('/')[1]}": c for k, c in ld("e2e_8b_full.pt").items() if k.startswith("bf16/")}
    base["batch4/seed0"], base["batch4/seed1"] = ld("e2e_8b_r3.pt")["batch4/bf16"], ld("e2e_8b_r3b.pt")["batch4/bf16"]
    rows, flips, worst, outside = 0, 0, 0.0, 0
    for key, c in ld("e2e_8b_r5.pt").items():
        name, tag = key.rsplit("/", 1)
        if tag not in ("t1", "t2", "t4") or name not in base:
            continue
        b = base[name]
        rows += int(b["logit"].numel())
        for i in (c["logit"] != b["logit"]).nonzero().flatten().tolist():
            flips += 1
            ids, vals = b["top_ids"][i].tolist(), b["top_values"][i].tolist()
            if int(c["logit"][i]) not in ids:
                outside += 1
                continue
            worst = max(worst, (vals[0] - vals[ids.index(int(c["logit"][i]))]) / _bf16_ulp(vals[0]))
    assert rows >= 390 and flips >= 10
    _REF_SELF_LEVELS[golden_dir] = (rows, flips, worst, outside)
    return _REF_SELF_LEVELS[golden_dir]


def _level_tie_bar(golden_dir):
    return REF_SELF_FACTOR * _ref_self_level_gap(golden_dir)[2]


def _ref_self_spread(golden_dir):
    """(mean, max, n) of the reference's bf16 score1 against ITSELF under other host thread counts, in bf16 ulps of the score: every
    ``*/t1 | t2 | t4`` case of tests/golden/e2e_8b_r5.pt against the recorded 8-thread pass of the same clips (e2e_8b_full.pt, e2e_8b_r3.pt,
    e2e_8b_r3b.pt) + the five pairs of e2e_8b_r4_self.pt."""
    if golden_dir in _REF_SELF:
        return _REF_SELF[golden_dir]
    ld = lambda f: torch.load(os.path.join(golden_dir, f), weights_only=True)["cases"]
    base = {f"one/{k.split('/')[1]}": c["score1"].float() for k, c in ld("e2e_8b_full.pt").items() if k.startswith("bf16/")}
    base["batch4/seed0"] = ld("e2e_8b_r3.pt")["batch4/bf16"]["score1"].float()
    base["batch4/seed1"] = ld("e2e_8b_r3b.pt")["batch4/bf16"]["score1"].float()
    d = []
    for key, c in ld("e2e_8b_r5.pt").items():
        name, tag = key.rsplit("/", 1)
        if tag in ("t1", "t2", "t4") and name in base:
            d += [_ulps(c["score1"].float()[i] - base[name][i], base[name][i]) for i in range(len(base[name]))]
    c4 = ld("e2e_8b_r4_self.pt")
    a, b = c4["batch4/seed0/t8"]["score1"].float(), c4["batch4/seed0/t4"]["score1"].float()
    d += [_ulps(a[i] - b[i], a[i]) for i in range(4)]
    d.append(_ulps(c4["alone/seed0/clip0/t1"]["score1"].float()[0] - c4["alone/seed0/clip0/t8"]["score1"].float()[0], a[0]))
    # (and the property that makes the per-clip plans of this path the right thing: alone == in batch, bit for bit)
    for seed in (0, 1):
        alone = torch.cat([c4[f"alone/seed{seed}/clip{i}/t8"]["score1"] for i in range(4)])
        assert torch.equal(alone, c4[f"batch4/seed{seed}/t8"]["score1"])
    assert len(d) >= 44
    _REF_SELF[golden_dir] = (sum(d) / len(d), max(d), len(d))
    return _REF_SELF[golden_dir]


def _bf16_ulp(x: float) -> float:
    return 2.0 ** (torch.tensor(abs(x)).clamp_min(1e-30).log2().floor().item() - 7)


def test_full_size_8b_matches_the_reference_golden(full_8b, golden_dir):
    """north_star's bar at the BASELINE configuration, against the reference itself (not the oracle).  Both checks are calibrated
    on what the reference's OWN two precisions do on these inputs (the fixture holds its bf16 and fp32 passes):
    * quality-level tokens.  With random weights the vocabulary logits are near-uniform: on 17 of the 50 answer rows the bf16
      reference's top two logits are within 2 bf16 ulps, and its bf16 and fp32 passes pick different tokens on 6 rows.  Bars:
      (a) per row, the HIP token is the bf16 reference's token or one of its top four whose logit the reference itself puts within
      LEVEL_TIE_ULPS = 4 bf16 ulps of its maximum (0.12 at |logit| ~ 5: the logit noise of 1-2 % of bf16 rounding noise in the
      final hidden state; two independent bf16-noisy evaluations differ by sqrt 2 of that); (b) against the FP32 reference the
      HIP tokens agree at least as often as the bf16 reference's tokens do (minus 4 rows: 40 against 44 measured).  The hard form of "levels bit-exact"
      is the planted-margin test below;
    * score1: the reference's bf16 number is itself ~0.011 away from its fp32 number at this depth (32 + 24 layers of bf16
      rounding on a random-weight model), so "within 1e-3 of the reference" is below the reference's own arithmetic noise.
      Bar: over the input seeds the HIP score is as close to the reference's FP32 score as the reference's bf16 score is
      (mean |hip - fp32| <= 1.5 x mean |ref bf16 - fp32| + one bf16 ulp), and per clip never further from the bf16 one than 1.3 x the
      largest distance of the reference to ITSELF under another host thread count (REF_SELF_FACTOR above: 44 pairs); the head's INPUT (hidden[:, -4]) is checked too: relative L2 to the fp32 reference <= 1.2 x the reference bf16's."""
    model, cfg, g = full_8b
    dev = model.device
    seeds = sorted({int(k.split("/")[1]) for k in g["cases"] if k.startswith("bf16/")})
    assert len(seeds) >= 3
    e_hip, e_ref, n_rows, n_tie, worst, worst_ulps, bad = [], [], 0, 0, 0.0, 0.0, []
    agree_hip, agree_ref, h_hip, h_ref = 0, 0, [], []
    for seed in seeds:
        r16, r32 = g["cases"][f"bf16/{seed}"], g["cases"][f"fp32/{seed}"]
        toks, pv, motion = _golden_inputs(cfg, seed, dev)
        model.img_context_token_id = toks["img_context_token_id"]
        out = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                    image_flags=torch.ones(8, 1, dtype=torch.long), labels=toks["labels"], motion_feature=motion)
        torch.cuda.synchronize()
        rows = r16["answer_rows"]
        assert torch.equal(out["label"].cpu()[rows], r16["label"])
        got = out["logit"].cpu()[rows]
        agree_hip += int((got == r32["logit"]).sum())
        agree_ref += int((r16["logit"] == r32["logit"]).sum())
        for i in (got != r16["logit"]).nonzero().flatten().tolist():
            ids, vals = r16["top_ids"][i].tolist(), r16["top_values"][i].tolist()
            if int(got[i]) not in ids:
                bad.append(f"seed {seed} row {i}: token {int(got[i])} is not among the reference's top four {ids}")
                continue
            gap = (vals[0] - vals[ids.index(int(got[i]))]) / _bf16_ulp(vals[0])
            print(f"seed {seed} answer row {i}: reference {int(r16['logit'][i])} vs hip {int(got[i])}, reference's own logit gap {gap:.2f} bf16 ulps"
                  f" (fp32 reference: {int(r32['logit'][i])})")
            if gap > _level_tie_bar(golden_dir):
                bad.append(f"seed {seed} row {i}: argmax differs beyond a near-tie of the reference ({gap:.2f} ulps)")
            n_tie += 1
        n_rows += len(rows)
        hip, b16, f32 = out["score1"].float().item(), r16["score1"].float().item(), r32["score1"].float().item()
        e_hip.append(abs(hip - f32)); e_ref.append(abs(b16 - f32)); worst = max(worst, abs(hip - b16)); worst_ulps = max(worst_ulps, _ulps(hip - b16, b16))
        hid = model.last_hidden_rows(1).float().cpu()[:, ::16]
        h32 = r32["hidden_m4_sub"].float()
        h_hip.append(float((hid - h32).norm() / h32.norm())); h_ref.append(float((r16["hidden_m4_sub"].float() - h32).norm() / h32.norm()))
        print(f"seed {seed}: score1 hip {hip:.6f} reference bf16 {b16:.6f} fp32 {f32:.6f}; hidden[:, -4] rel L2 vs fp32 reference: hip {h_hip[-1]:.4f}, "
              f"reference bf16 {h_ref[-1]:.4f}")
    m_hip, m_ref = sum(e_hip) / len(seeds), sum(e_ref) / len(seeds)
    print(f"full size: level tokens {n_rows - n_tie}/{n_rows} identical to the bf16 reference ({n_tie} near-ties); agreement with the fp32 "
          f"reference: hip {agree_hip}/{n_rows}, bf16 reference {agree_ref}/{n_rows}; mean |hip - fp32| {m_hip:.5f}, "
          f"mean |reference bf16 - fp32| {m_ref:.5f}, max |hip - reference bf16| {worst:.5f}")
    assert not bad, bad
    assert n_tie <= n_rows // 4
    # 50 rows: the count moves by +-3 with the kernel choice alone (44 with the pipelined ViT attention, 40 with the default one, against
    # 44 for the reference's own bf16 pass) - every disagreement is a near-tie (asserted above); a wrong kernel lands far below this
    assert agree_hip >= agree_ref - 4
    assert m_hip <= 1.5 * m_ref + 2.0 ** -8
    assert worst_ulps <= REF_SELF_FACTOR * _ref_self_spread(golden_dir)[1], worst_ulps
    # the score head's input, hidden_states[-1][:, -4, :] (a 256-value subsample is recorded): relative L2 distance to the fp32
    # reference no larger than 1.2 x the reference bf16 pass's own
    assert max(h_hip) <= 1.2 * max(h_ref) and sum(h_hip) <= 1.2 * sum(h_ref), (h_hip, h_ref)


def test_full_size_8b_planted_margin_levels_are_bit_exact(full_8b, golden_dir):
    """The golden's weights with five 'quality level' rows of the lm-head scaled by 8 (ids chosen by the generator so that one
    of them wins every answer row by >= 0.75 sigma of that row's vocabulary logits): a trained model's situation - a clear
    winner - where 'quality levels bit-exact' is a hard assert with NO tie exemption."""
    model, cfg, g = full_8b
    key = [k for k in g["cases"] if k.startswith("planted/")][0]
    rec = g["cases"][key]
    seed = rec["seed"]
    w = model.language_model.output.weight
    keep = w.data[rec["level_ids"]].clone()
    try:
        w.data[rec["level_ids"]] = keep * g["plant_scale"]
        model._invalidate()
        toks, pv, motion = _golden_inputs(cfg, seed, model.device)
        model.img_context_token_id = toks["img_context_token_id"]
        out = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                    image_flags=torch.ones(8, 1, dtype=torch.long), labels=toks["labels"], motion_feature=motion)
        got = out["logit"].cpu()[rec["answer_rows"]]
        print("planted levels: hip", got.tolist(), "reference", rec["logit"].tolist(), "margins (sigma)", [round(float(x), 2) for x in rec["margin_sigma"]])
        assert torch.equal(got, rec["logit"])
        assert len(set(got.tolist())) >= 3 and set(got.tolist()) <= set(rec["level_ids"])
        # (the score does not see the lm-head: this is clip one/201 again, which the reference itself moves by 3-5 ulps with its thread count)
        d = _ulps(out["score1"].float().item() - rec["score1"].float().item(), rec["score1"].float().item())
        print(f"planted levels: score1 {d:.1f} bf16 ulps from the reference's")
        assert d <= REF_SELF_FACTOR * _ref_self_spread(golden_dir)[1], d
    finally:
        w.data[rec["level_ids"]] = keep
        model._invalidate()


def test_full_size_8b_fp8_mode_keeps_the_planted_margin_levels(full_8b, golden_dir):
    """BASELINE config 5's arithmetic (set_precision('fp8'): e4m3 InternLM2 linears, per-row / per-channel scales) at FULL DEPTH against the
    bf16 REFERENCE, on the thing the mode is for - quality levels: with the planted-margin lm-head rows (winners by
    0.79-1.85 sigma in the reference's own pass) a usable fp8 mode picks the reference's token on EVERY answer row.  The score
    is a regression on a 4096-wide hidden state and drifts with 32 layers of e4m3 rounding: measured at TASK level (SRCC / PLCC over the 37
    pinned clips, second half of this test) and found not usable - the mode is opt-in, experimental and never the headline."""
    model, cfg, g = full_8b
    key = [k for k in g["cases"] if k.startswith("planted/")][0]
    rec = g["cases"][key]
    w = model.language_model.output.weight
    keep = w.data[rec["level_ids"]].clone()
    try:
        w.data[rec["level_ids"]] = keep * g["plant_scale"]
        model._invalidate()
        model.set_precision("fp8")
        toks, pv, motion = _golden_inputs(cfg, rec["seed"], model.device)
        model.img_context_token_id = toks["img_context_token_id"]
        out = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                    image_flags=torch.ones(8, 1, dtype=torch.long), labels=toks["labels"], motion_feature=motion)
        torch.cuda.synchronize()
        got = out["logit"].cpu()[rec["answer_rows"]]
        hip, ref = out["score1"].float().item(), rec["score1"].float().item()
        print("fp8 mode, planted levels: hip", got.tolist(), "reference (bf16)", rec["logit"].tolist(), "margins (sigma)",
              [round(float(x), 2) for x in rec["margin_sigma"]], f"; score1 fp8 mode {hip:.6f} reference bf16 {ref:.6f} |d| {abs(hip - ref):.4f} "
              f"= {abs(hip - ref) / _bf16_ulp(ref):.1f} bf16 ulps")
        # MEASURED (round 5): with the reference's score rounding in the attention all ten rows are the reference's; with fp32 scores (the
        # default) the row the reference wins by the smallest margin (0.79 sigma) flips.  VERDICT r4 item 6's criterion - a usable fp8 mode
        # keeps them all - is therefore NOT met: the mode is documented as an experiment (README).  What is asserted
        # is what holds under both numerics: every level the reference wins by >= 0.85 sigma, and at most one flip in all.
        flips = (got != rec["logit"]).nonzero().flatten().tolist()
        print("fp8 mode: flipped rows", flips, "their margins (sigma)", [round(float(rec["margin_sigma"][i]), 2) for i in flips])
        assert len(flips) <= 1 and all(float(rec["margin_sigma"][i]) < 0.85 for i in flips), flips
        # (no bar on this one score: until round 5 a `<= 0.35` on a [0, 1] score stood here, which said nothing.  What the mode does to SCORES is
        # measured at task level below - and it is not usable.)
        # TASK LEVEL (round 6): SRCC / PLCC of the fp8-mode scores against the reference's bf16 scores over the 37 pinned clips -
        # with the plain (un-planted) lm-head the scores do not depend on it.  MEASURED: SRCC 0.71-0.73, PLCC 0.75-0.78, mean |d| 17 bf16 ulps
        # (profiles/r6_parity_stats.txt), against 0.985 / 0.995 / 2.8 for the bf16 path and 0.984 / 0.9975 / 2.56 for the reference against itself:
        # on this evidence (iid random weights, 32 layers of e4m3 rounding) the mode's scores are NOT usable, and README / DESIGN.md say so.  The
        # assert below is a regression guard on the measured level, not a usability claim.
        s_hip, s_ref = [], []
        for B, seed, r16 in _pinned_cases(g, golden_dir):
            toks = synth.canonical_tokens(cfg, B, 8, seed=seed)
            model.img_context_token_id = toks["img_context_token_id"]
            o = model(mos=None, pixel_values=synth.synthetic_frames(B * 8, 448, seed=seed).to(model.device), input_ids=toks["input_ids"],
                      attention_mask=toks["attention_mask"], image_flags=torch.ones(B * 8, 1, dtype=torch.long), labels=toks["labels"],
                      motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(model.device))
            s_hip += o["score1"].float().cpu().tolist(); s_ref += r16["score1"].float().tolist()
        c8 = _corr(s_hip, s_ref)
        d8 = sum(_ulps(a - b, b) for a, b in zip(s_hip, s_ref)) / len(s_ref)
        print(f"fp8 mode, task level over {len(s_ref)} clips vs the reference's bf16 scores: SRCC {c8[0]:.4f} PLCC {c8[1]:.4f} KRCC {c8[2]:.4f}, mean |d| {d8:.1f} bf16 ulps "
              f"- NOT usable as scores (the bf16 path: 0.985 / 0.995; the reference against itself: 0.984 / 0.9975)")
        assert len(s_ref) == 37 and c8[0] >= 0.55 and c8[1] >= 0.6, c8
    finally:
        model.set_precision("bf16")
        w.data[rec["level_ids"]] = keep
        model._invalidate()


def test_full_size_8b_properties(full_8b):
    """BASELINE.json configs[1] exactly (4 clips x 8 frames x 448 px, N = 2177, full depth): size-independent properties of
    the product path - determinism, batch invariance (four clips scored together == each scored alone, bit for bit, in the DEFAULT
    mode), frame-DP equivalence through score_clips_dp, answer rows filled."""
    from aigv_assessor_amd.dist_utils import score_clips_dp
    model, cfg, _g = full_8b
    dev = model.device
    B, T = 4, 8
    toks = synth.canonical_tokens(cfg, B, T, seed=3)
    model.img_context_token_id = toks["img_context_token_id"]
    pv = synth.synthetic_frames(B * T, 448, seed=3, device=dev)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=3, device=dev)
    flags = torch.ones(B * T, 1, dtype=torch.long)

    def run(sl_c, sl_f):
        return model(pixel_values=pv[sl_f], input_ids=toks["input_ids"][sl_c], attention_mask=toks["attention_mask"][sl_c],
                     image_flags=flags[sl_f], labels=toks["labels"][sl_c], motion_feature=motion[sl_c])
    both = run(slice(0, B), slice(0, B * T))
    again = run(slice(0, B), slice(0, B * T))
    assert torch.equal(both["score1"], again["score1"]) and torch.equal(both["logit"], again["logit"])      # deterministic
    n1 = toks["input_ids"].shape[1] - 1
    assert n1 == 2176
    # Batch invariance in the DEFAULT mode (round 4).  The GEMM dispatch is planned per clip / per frame (csrc/api.hip
    # struct RowPlan): a row's kernel form and K split follow from its place in its own sequence, never from the batch - so a clip scored
    # alone gives the bits it gives inside the batch (until round 3: only to bf16 noise, 0.027 on this very clip).  Modes 1 / 2 (every
    # row on one tile kernel in full K) are batch-invariant as well and stay as test aliases.
    for mode in (-1, 1):
        model.set_gemm_mode(mode)
        try:
            ref = both if mode == -1 else run(slice(0, B), slice(0, B * T))
            for b in range(B):
                one = run(slice(b, b + 1), slice(8 * b, 8 * b + 8))
                if mode == -1:
                    print(f"clip {b}: alone {one['score1'].item():.4f} vs in batch {ref['score1'][b].item():.4f}")
                assert torch.equal(one["score1"], ref["score1"][b:b + 1]), (mode, b)
                assert torch.equal(one["logit"], ref["logit"][b * n1:(b + 1) * n1]), (mode, b)
            two = run(slice(1, 3), slice(8, 24))                                  # a different batch composition
            assert torch.equal(two["score1"], ref["score1"][1:3]) and torch.equal(two["logit"], ref["logit"][n1:3 * n1]), mode
        finally:
            model.set_gemm_mode(-1)
    want = (toks["labels"][:, 1:] != -100).reshape(-1)
    lg = both["logit"].cpu()
    assert (lg[want] >= 0).all() and (lg[want] < cfg.llm_config.vocab_size).all() and (lg[~want] == -1).all()
    assert torch.isfinite(both["score1"].float()).all() and (both["score1"].float() >= 0).all()      # ReLU head
    dp = score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion)
    assert torch.equal(dp["score1"], both["score1"]) and torch.equal(dp["logit"], both["logit"])


def test_full_size_8b_conditioned_weights_against_the_reference(full_8b, golden_dir):
    """the same seeded weights made to look like a trained checkpoint (synth.condition_state_dict: InternViT layer scales
    x 0.1, InternLM2 wo / w2 x 1 / sqrt(2 L)), 32 clips recorded from the imported reference in bf16 (16 of them also under 4 (/ 2 / 1) host threads), all of
    them in fp32 (tests/golden/make_golden_8b_conditioned.py).  The question was whether the reference is then stable against itself to <= 1 bf16
    ulp, so that HIP could be held to a hard per-clip bar.  MEASURED: it is not - its passes under other thread counts sit 3.1
    bf16 ulps (mean; max 10) from its 8-thread pass, and its bf16 pass 4.7 (max 11.9) from its OWN fp32 pass.  In absolute terms the noise is
    what it is on the iid weights (~0.01); the scores are smaller here (0.2-0.7), so it counts more ulps.  Asserted, all read from the fixture:
      * as close to the reference's fp32 scores as the reference's own bf16 pass is (mean, factor 1.3) - the bar of the iid test;
      * no clip farther from the reference's bf16 score than 1.3 x the largest distance the reference shows against itself (thread counts, fp32);
      * level tokens identical up to the reference's own near-ties.
    REPORTED, not asserted: the pooled mean |hip - ref bf16| (6.0 ulps) against the thread-count spread (3.1).  Passes of ONE implementation that
    differ in nothing but GEMM blocking share most of their rounding decisions - that spread is a correlated lower bound; two evaluations that
    each sit ~5 ulps from the fp32 value (the reference's bf16 pass: 4.7; HIP: 5.0) are expected ~6-7 apart.  The hidden-state bars below are the
    robust form of the same comparison (HIP vs ref bf16 0.0450 against a thread-count spread of 0.0376 and a bf16-vs-fp32 distance of 0.0409)."""
    path = os.path.join(golden_dir, "e2e_8b_conditioned.pt")
    if not os.path.exists(path):
        pytest.skip("tests/golden/e2e_8b_conditioned.pt not generated")
    model, cfg, g = full_8b
    c = torch.load(path, weights_only=True)
    assert c["w_seed"] == g["w_seed"] and c["overrides"] == g["overrides"] and c["conditioned"] is True
    cases = c["cases"]
    seeds = sorted({int(k.split("/")[1][4:]) for k in cases if k.endswith("/bf16/t8")})
    dev = model.device
    L = cfg.llm_config.num_hidden_layers
    down = 1.0 / (2.0 * L) ** 0.5
    keep = {}
    try:
        with torch.no_grad():     # synth.condition_state_dict's arithmetic on the device copies (fp32 multiply, one rounding back): the same bits
            for k, p_ in model.named_parameters():
                f = 0.1 if (k.startswith("vision_model.encoder.layers.") and (k.endswith(".ls1") or k.endswith(".ls2"))) else \
                    down if (k.startswith("language_model.model.layers.") and (k.endswith("attention.wo.weight") or k.endswith("feed_forward.w2.weight"))) else None
                if f is not None:
                    keep[k] = p_.data.clone()
                    p_.data.copy_((p_.data.float() * f).to(p_.dtype))
        probe = synth.condition_state_dict({"vision_model.encoder.layers.0.ls1": keep["vision_model.encoder.layers.0.ls1"].cpu().clone()}, cfg)
        assert torch.equal(probe["vision_model.encoder.layers.0.ls1"], dict(model.named_parameters())["vision_model.encoder.layers.0.ls1"].data.cpu())
        model._invalidate()
        d_hip, d_self, d32_hip, d32_ref, lev_rows = [], [], [], [], 0
        lev_hip, lev_ref = [0, 0, 0, 0], [0, 0, 0, 0]             # flips, flips to a token outside the 8-thread pass's top four, flips beyond the near-tie bar, rows
        h_hip, h_self, h32_hip, h32_ref = [], [], [], []          # the same four distances on hidden[:, -4] (relative L2 over its 4096 coordinates)
        s_hip, s_ref16, s_hip_on32, s_ref16_on32, s_ref32 = [], [], [], [], []   # the scores themselves, for the task-level statistics (SRCC / PLCC, stage2_eval.py:652-688)
        for seed in seeds:
            r8 = cases[f"batch4/seed{seed}/bf16/t8"]
            toks = synth.canonical_tokens(cfg, 4, 8, seed=seed)
            model.img_context_token_id = toks["img_context_token_id"]
            out = model(mos=None, pixel_values=synth.synthetic_frames(32, 448, seed=seed).to(dev), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                        image_flags=torch.ones(32, 1, dtype=torch.long), labels=toks["labels"], motion_feature=synth.synthetic_motion(4, cfg.motion_dim, seed=seed).to(dev))
            torch.cuda.synchronize()
            hip, want = out["score1"].float().cpu(), r8["score1"].float()
            hid = model.last_hidden_rows(4).cpu()
            d_hip += [_ulps(hip[i] - want[i], want[i]) for i in range(4)]
            h_hip += _rel_l2(hid, r8["hidden_m4"])
            s_hip += hip.tolist()
            s_ref16 += want.tolist()
            for t in (1, 2, 4):
                o = cases.get(f"batch4/seed{seed}/bf16/t{t}")
                if o is not None:
                    d_self += [_ulps(o["score1"].float()[i] - want[i], want[i]) for i in range(4)]
                    h_self += _rel_l2(o["hidden_m4"], r8["hidden_m4"])
            r32 = cases.get(f"batch4/seed{seed}/fp32/t8")
            if r32 is not None:
                w32 = r32["score1"].float()
                d32_hip += [_ulps(hip[i] - w32[i], want[i]) for i in range(4)]
                d32_ref += [_ulps(want[i] - w32[i], want[i]) for i in range(4)]
                h32_hip += _rel_l2(hid, r32["hidden_m4"])
                h32_ref += _rel_l2(r8["hidden_m4"], r32["hidden_m4"])
                s_hip_on32 += hip.tolist()
                s_ref16_on32 += want.tolist()
                s_ref32 += w32.tolist()
            # level tokens: on these weights the vocabulary logits are near-uniform and the reference flips 8.5 % of its own level tokens with the thread
            # count, some to tokens outside its recorded top four - counted for both sides, compared below
            tie = _level_tie_bar(golden_dir)
            for got_ids, is_hip in [(out["logit"].cpu()[r8["answer_rows"]], True)] + [(cases[f"batch4/seed{seed}/bf16/t{t}"]["logit"], False) for t in (1, 2, 4)
                                                                                       if f"batch4/seed{seed}/bf16/t{t}" in cases]:
                flips, outside, beyond = _level_flip_stats(got_ids, r8, tie)
                tgt = lev_hip if is_hip else lev_ref
                tgt[0] += flips; tgt[1] += outside; tgt[2] += beyond; tgt[3] += int(r8["logit"].numel())
            lev_rows += int(r8["answer_rows"].numel())
        m_hip, m_self = sum(d_hip) / len(d_hip), sum(d_self) / len(d_self)
        a, b = sum(d32_hip) / len(d32_hip), sum(d32_ref) / len(d32_ref)
        print(f"conditioned weights, {len(d_hip)} clips: |hip - ref bf16| mean {m_hip:.2f} max {max(d_hip):.1f} bf16 ulps {[round(x, 1) for x in d_hip]}; the reference against itself "
              f"({len(d_self)} pairs: other host thread counts vs 8) mean {m_self:.2f} max {max(d_self):.1f}; level tokens flipped: hip {lev_hip[0]}/{lev_hip[3]} "
              f"({lev_hip[1]} outside the reference's top four, {lev_hip[2]} beyond the near-tie bar), the reference against itself {lev_ref[0]}/{lev_ref[3]} ({lev_ref[1]}, {lev_ref[2]})")
        print(f"conditioned weights, against the reference's fp32 scores ({len(d32_hip)} clips): hip mean {a:.2f} max {max(d32_hip):.1f} bf16 ulps, the reference's own bf16 pass "
              f"{b:.2f} max {max(d32_ref):.1f}")
        mean = lambda v: sum(v) / len(v)
        print(f"conditioned weights, hidden[:, -4] relative L2: hip vs ref bf16 {mean(h_hip):.4f}, the reference vs itself {mean(h_self):.4f}; hip vs ref fp32 {mean(h32_hip):.4f}, "
              f"the reference's bf16 pass vs its fp32 pass {mean(h32_ref):.4f}")
        assert len(d_hip) >= 8 and len(d_self) >= 8 and len(d32_hip) >= 8
        # task level (the reference's own quality measure): on the 32 recorded clips HIP ranks like the reference - SRCC 0.9875 / PLCC 0.9917 against its bf16 scores,
        # 0.9911 / 0.9936 against its fp32 scores, where the reference's bf16 pass reaches 0.9944 / 0.9951 against its own fp32 pass
        c16, c32, cref = _corr(s_hip, s_ref16), _corr(s_hip_on32, s_ref32), _corr(s_ref16_on32, s_ref32)
        print(f"conditioned weights, task level over {len(s_hip)} clips (SRCC / PLCC / KRCC): hip vs ref bf16 {c16[0]:.4f} / {c16[1]:.4f} / {c16[2]:.4f}; hip vs ref fp32 "
              f"{c32[0]:.4f} / {c32[1]:.4f} / {c32[2]:.4f}; the reference's bf16 pass vs its fp32 pass {cref[0]:.4f} / {cref[1]:.4f} / {cref[2]:.4f}")
        if len(s_hip) >= 24:
            assert c16[0] >= 0.97 and c16[1] >= 0.98, c16
            assert c32[0] >= cref[0] - 0.015 and c32[1] >= cref[1] - 0.01, (c32, cref)
        # the 4096-wide hidden state the score head reads - a vector norm, far less noisy than the scalar: as close to the fp32 computation as the reference's bf16
        # pass is, and no farther from the reference's bf16 pass than 1.3 x the larger of the reference's own two distances
        assert mean(h32_hip) <= REF_SELF_FACTOR * mean(h32_ref), (mean(h32_hip), mean(h32_ref))
        assert mean(h_hip) <= REF_SELF_FACTOR * max(mean(h_self), mean(h32_ref)), (mean(h_hip), mean(h_self), mean(h32_ref))
        assert a <= REF_SELF_FACTOR * b, (a, b)                                                    # as close to the fp32 computation as the reference's bf16 pass
        assert max(d_hip) <= REF_SELF_FACTOR * max(max(d_self), max(d32_ref)), (d_hip, d_self, d32_ref)    # no clip beyond the reference's own largest distance
        # level tokens: no more flips (in all, and of the two kinds a near-tie rule would reject) than 1.3 x the reference's own rate against itself, + 1 row
        for i in range(3):
            assert lev_hip[i] <= REF_SELF_FACTOR * lev_ref[i] / lev_ref[3] * lev_hip[3] + 1, (lev_hip, lev_ref)
    finally:
        with torch.no_grad():
            for k, p_ in model.named_parameters():
                if k in keep:
                    p_.data.copy_(keep[k])
        model._invalidate()


def test_full_size_8b_eight_clips_per_gpu_share_of_configs_3_and_5(full_8b):
    """The per-GPU share of BASELINE config 5 (64 clips x 8 frames on 8 GPUs = 8 clips per GPU; config 3's share, 4 clips, is the headline
    itself) at full depth on the one card: 8 clips x 8 frames x 448 px = 17408 token rows in one pass.  Size-independent
    properties in bf16 AND in the e4m3 mode of config 5: deterministic; a clip scored alone and a pair scored together give the bits they
    give inside the batch of eight; the frame / clip data-parallel scorer (one rank) equals the plain forward; the first four clips score
    as in a batch of four (the headline shape).  The N > 1 half of both configs stays unmeasured on hardware (no 8-GPU node for the builder)."""
    from aigv_assessor_amd.dist_utils import score_clips_dp
    model, cfg, _g = full_8b
    dev = model.device
    B, T = 8, 8
    toks = synth.canonical_tokens(cfg, B, T, seed=5)
    model.img_context_token_id = toks["img_context_token_id"]
    pv = synth.synthetic_frames(B * T, 448, seed=5, device=dev)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=5, device=dev)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    n1 = toks["input_ids"].shape[1] - 1

    def run(c0, c1):
        return model(pixel_values=pv[c0 * T:c1 * T], input_ids=toks["input_ids"][c0:c1], attention_mask=toks["attention_mask"][c0:c1],
                     image_flags=flags[c0 * T:c1 * T], labels=toks["labels"][c0:c1], motion_feature=motion[c0:c1])
    try:
        for precision in ("bf16", "fp8"):
            model.set_precision(precision)
            whole = run(0, B)
            again = run(0, B)
            torch.cuda.synchronize()
            assert torch.equal(whole["score1"], again["score1"]) and torch.equal(whole["logit"], again["logit"]), precision      # deterministic
            assert torch.isfinite(whole["score1"].float()).all() and len(set(whole["score1"].float().tolist())) >= 4, precision
            for c0, c1 in ((2, 3), (7, 8), (4, 6), (0, 4)):           # alone, alone, a pair, the headline batch of four
                part = run(c0, c1)
                assert torch.equal(part["score1"], whole["score1"][c0:c1]), (precision, c0, c1, part["score1"], whole["score1"][c0:c1])
                assert torch.equal(part["logit"], whole["logit"][c0 * n1:c1 * n1]), (precision, c0, c1)
            dp = score_clips_dp(model, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion)
            assert torch.equal(dp["score1"], whole["score1"]) and torch.equal(dp["logit"], whole["logit"]), precision
            print(f"8 clips per GPU, {precision}: score1 {[round(x, 4) for x in whole['score1'].float().tolist()]}")
    finally:
        model.set_precision("bf16")


# ---------------------------------------------------------------------------------------------------------
# Round 3: the reference's recorded outputs (tests/golden/make_golden_8b_r3.py -> e2e_8b_r3.pt) for (1) the batch bench.py times,
# (2) a stage-1 pass on a 16-frame clip, (3) generate() through the reference's own KV-cache path with planted-margin level rows.
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_r3(golden_dir):
    return torch.load(os.path.join(golden_dir, "e2e_8b_r3.pt"), weights_only=True)


def _level_rows_ok(got, r16, tag, tie_ulps=LEVEL_TIE_ULPS):
    """Level tokens against the reference's bf16 pass: identical, or a token the reference itself puts within ``tie_ulps`` of its
    maximum (its recorded top four).  Returns the number of such near-ties; raises on any other difference."""
    bad, n_tie = [], 0
    for i in (got != r16["logit"]).nonzero().flatten().tolist():
        ids, vals = r16["top_ids"][i].tolist(), r16["top_values"][i].tolist()
        if int(got[i]) not in ids:
            bad.append(f"{tag} row {i}: token {int(got[i])} is not among the reference's top four {ids}")
            continue
        gap = (vals[0] - vals[ids.index(int(got[i]))]) / _bf16_ulp(vals[0])
        print(f"{tag} answer row {i}: reference {int(r16['logit'][i])} vs hip {int(got[i])}, reference's own logit gap {gap:.2f} bf16 ulps")
        if gap > tie_ulps:
            bad.append(f"{tag} row {i}: argmax differs beyond a near-tie of the reference ({gap:.2f} ulps, bar {tie_ulps:.1f})")
        n_tie += 1
    assert not bad, bad
    return n_tie


@pytest.mark.parametrize("fixture,input_seed", [("e2e_8b_r3.pt", 0), ("e2e_8b_r3b.pt", 1)])
def test_full_size_8b_benched_batch_matches_the_reference(full_8b, golden_dir, fixture, input_seed):
    """BASELINE.json configs[1] AS bench.py TIMES IT - 4 clips x 8 frames x 448 px in one call (8704-row GEMM bands, split-K tails, the
    column-group tile order of the large InternLM2 matrices), bench.py's own inputs (seed 0 tokens / frames, motion_feature as an input) -
    against the imported reference's bf16 and fp32 passes over the same batch; and a second batch of the same shape (inputs of seed 1,
    tests/golden/make_golden_8b_r3b.py), so that the statistics do not rest on one batch.  Same bars as the one-clip test above."""
    model, cfg, g = full_8b
    golden_r3 = torch.load(os.path.join(golden_dir, fixture), weights_only=True)
    assert golden_r3["w_seed"] == g["w_seed"] and golden_r3["overrides"] == g["overrides"]
    r16, r32 = golden_r3["cases"]["batch4/bf16"], golden_r3["cases"]["batch4/fp32"]
    B, T, seed = r16["B"], r16["T"], r16["seed"]
    assert (B, T, seed) == (4, 8, input_seed)
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    model.img_context_token_id = toks["img_context_token_id"]
    dev = model.device
    out = model(mos=None, pixel_values=synth.synthetic_frames(B * T, 448, seed=seed).to(dev), input_ids=toks["input_ids"],
                attention_mask=toks["attention_mask"], image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=toks["labels"],
                motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(dev))
    torch.cuda.synchronize()
    rows = r16["answer_rows"]
    assert torch.equal(out["label"].cpu()[rows], r16["label"]) and r16["n_rows"] == out["label"].numel()
    got = out["logit"].cpu()[rows]
    n_tie = _level_rows_ok(got, r16, "batch of 4", _level_tie_bar(golden_dir))
    agree_hip, agree_ref = int((got == r32["logit"]).sum()), int((r16["logit"] == r32["logit"]).sum())
    hip, b16, f32 = out["score1"].float().cpu(), r16["score1"].float(), r32["score1"].float()
    m_hip, m_ref, worst = float((hip - f32).abs().mean()), float((b16 - f32).abs().mean()), float((hip - b16).abs().max())
    hid, h32 = model.last_hidden_rows(B).float().cpu(), r32["hidden_m4"].float()
    h_hip = ((hid - h32).norm(dim=1) / h32.norm(dim=1)).tolist()
    h_ref = ((r16["hidden_m4"].float() - h32).norm(dim=1) / h32.norm(dim=1)).tolist()
    print(f"benched batch: score1 hip {hip.tolist()} reference bf16 {b16.tolist()} fp32 {f32.tolist()}; level tokens {len(rows) - n_tie}/{len(rows)} identical "
          f"({n_tie} near-ties), agreement with fp32: hip {agree_hip}, reference bf16 {agree_ref}; mean |hip - fp32| {m_hip:.5f}, mean |ref bf16 - fp32| "
          f"{m_ref:.5f}, max |hip - ref bf16| {worst:.5f}; hidden[:, -4] rel L2 vs fp32: hip {[round(x, 4) for x in h_hip]} reference bf16 {[round(x, 4) for x in h_ref]}")
    assert n_tie <= len(rows) // 4
    assert agree_hip >= agree_ref - 4
    # (the anchor on the reference's FP32 scores is a pooled statistic - four clips are too few for a ratio of two means: on this batch the
    #  reference's own bf16 pass happens to sit 1.4 ulps from its fp32 pass, on the other 2.9 - and lives in the pooled test below)
    for b in range(B):
        assert _ulps(hip[b] - b16[b], b16[b]) <= REF_SELF_FACTOR * _ref_self_spread(golden_dir)[1], (b, float(hip[b]), float(b16[b]))
    assert max(h_hip) <= 1.2 * max(h_ref) and sum(h_hip) <= 1.2 * sum(h_ref)


def test_full_size_8b_pooled_score_distance_is_the_references_own_spread(full_8b, golden_dir):
    """All 37 clips the imported reference was recorded on (five one-clip seeds, two batches of four of round 3, six of round 5) in one
    statistic: the MEAN distance of the HIP scores to the reference's bf16 scores must not exceed 1.3 x the mean distance of the reference to
    ITSELF when nothing but the host's thread count changes (44 pairs, _ref_self_spread), and no clip 1.3 x its maximum - i.e. the HIP path is
    one more evaluation of the reference's arithmetic, as far from the recorded one as the reference on another machine would be.  The six new
    batches also check their level tokens (every disagreement a near-tie of the reference's own logits).  north_star's literal 1e-3 lies
    below that spread and is reported, not asserted: the share of clips with the identical bf16 score is printed."""
    model, cfg, g = full_8b
    dev = model.device
    self_mean, self_max, self_n = _ref_self_spread(golden_dir)
    cases = [(1, s, g["cases"][f"bf16/{s}"], False, g["cases"][f"fp32/{s}"]) for s in sorted({int(k.split("/")[1]) for k in g["cases"] if k.startswith("bf16/")})]
    for f in ("e2e_8b_r3.pt", "e2e_8b_r3b.pt"):
        r = torch.load(os.path.join(golden_dir, f), weights_only=True)["cases"]
        cases.append((r["batch4/bf16"]["B"], r["batch4/bf16"]["seed"], r["batch4/bf16"], False, r["batch4/fp32"]))
    c5 = torch.load(os.path.join(golden_dir, "e2e_8b_r5.pt"), weights_only=True)["cases"]
    for seed in range(2, 8):
        cases.append((4, seed, c5[f"batch4/seed{seed}/bf16"], True, c5.get(f"batch4/seed{seed}/fp32")))
    d, n_tie, n_rows, d32_hip, d32_ref, s_hip, s_ref, s_hip32, s_ref32 = [], 0, 0, [], [], [], [], [], []
    h_hip, h32_hip, h32_ref = [], [], []
    for B, seed, rec, check_levels_too, rec32 in cases:
        toks = synth.canonical_tokens(cfg, B, 8, seed=seed)
        model.img_context_token_id = toks["img_context_token_id"]
        out = model(mos=None, pixel_values=synth.synthetic_frames(B * 8, 448, seed=seed).to(dev), input_ids=toks["input_ids"],
                    attention_mask=toks["attention_mask"], image_flags=torch.ones(B * 8, 1, dtype=torch.long), labels=toks["labels"],
                    motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(dev))
        torch.cuda.synchronize()
        want = rec["score1"].float()
        hip = out["score1"].float().cpu()
        d += [_ulps(hip[i] - want[i], want[i]) for i in range(B)]
        s_hip += hip.tolist(); s_ref += want.tolist()
        if "hidden_m4" in rec:          # the 4096-wide hidden state the score head reads: relative L2 against the reference's bf16 / fp32 records of it
            hid = model.last_hidden_rows(B).cpu()
            h_hip += _rel_l2(hid, rec["hidden_m4"])
            if rec32 is not None and "hidden_m4" in rec32:
                h32_hip += _rel_l2(hid, rec32["hidden_m4"]); h32_ref += _rel_l2(rec["hidden_m4"], rec32["hidden_m4"])
        if rec32 is not None:     # the reference's fp32 pass of the same clip: how far each bf16-level evaluation sits from the fp32 computation
            w32 = rec32["score1"].float()
            s_hip32 += hip.tolist(); s_ref32 += w32.tolist()
            d32_hip += [_ulps(hip[i] - w32[i], want[i]) for i in range(B)]
            d32_ref += [_ulps(want[i] - w32[i], want[i]) for i in range(B)]
        if check_levels_too:
            n_tie += _level_rows_ok(out["logit"].cpu()[rec["answer_rows"]], rec, f"batch of 4, input seed {seed}", _level_tie_bar(golden_dir))
            n_rows += int(rec["answer_rows"].numel())
    mean, worst = sum(d) / len(d), max(d)
    print(f"{len(d)} reference-pinned clips: |hip - ref bf16| mean {mean:.2f} max {worst:.1f} bf16 ulps, identical bf16 score on {sum(1 for x in d if x == 0)}/{len(d)}, "
          f"within 1e-3 on {sum(1 for x in d if x < 0.3)}/{len(d)}; the reference against itself ({self_n} pairs, host threads): mean {self_mean:.2f} max {self_max:.1f}; "
          f"level tokens of the six round-5 batches: {n_rows - n_tie}/{n_rows} identical ({n_tie} near-ties of the reference's own logits)")
    m32_hip, m32_ref = sum(d32_hip) / len(d32_hip), sum(d32_ref) / len(d32_ref)
    print(f"against the reference's FP32 scores ({len(d32_hip)} clips with an fp32 record): hip mean {m32_hip:.2f} bf16 ulps, the reference's own bf16 pass {m32_ref:.2f}")
    assert len(d) == 37 and len(d32_hip) >= 13
    assert mean <= REF_SELF_FACTOR * self_mean, (mean, self_mean)
    assert worst <= REF_SELF_FACTOR * self_max, (worst, self_max)
    assert m32_hip <= REF_SELF_FACTOR * m32_ref, (m32_hip, m32_ref)       # as close to the fp32 computation as the reference's own bf16 pass is
    assert n_tie <= n_rows // 4
    # TASK-LEVEL agreement (round 6): SRCC / PLCC / KRCC - the reference's own quality metric (stage2_eval.py:652-688) - of the
    # HIP scores against the reference's over the 37 clips.  Bars read from the fixtures: what the reference reaches against ITSELF when only
    # the host thread count changes (its 1 / 2 / 4-thread passes of the 13 re-scored clips against its 8-thread pass, pooled: 39 pairs), and
    # its own bf16 pass against its fp32 pass - minus 0.01.
    hip16, hip32 = _corr(s_hip, s_ref), _corr(s_hip32, s_ref32)
    self16, self32 = _ref_self_corr(golden_dir), _corr(_ref16_on_fp32_clips(cases), s_ref32)
    print(f"task level over {len(s_hip)} clips: hip vs ref bf16 SRCC {hip16[0]:.4f} PLCC {hip16[1]:.4f} KRCC {hip16[2]:.4f};  reference vs itself (39 pairs) "
          f"SRCC {self16[0]:.4f} PLCC {self16[1]:.4f} KRCC {self16[2]:.4f};  hip vs ref fp32 SRCC {hip32[0]:.4f} PLCC {hip32[1]:.4f};  ref bf16 vs ref fp32 "
          f"SRCC {self32[0]:.4f} PLCC {self32[1]:.4f}")
    # hidden[:, -4] (round 6): relative L2 over 4096 coordinates - the reference against itself under other thread counts: 0.0356 (24 clips, computed below from the
    # fixtures), its bf16 pass against its fp32 pass 0.0329; measured for this path 0.0390 / 0.0331 (profiles/r6_hidden_distance.txt)
    h_self = []
    for key, c in c5.items():
        name, tag = key.rsplit("/", 1)
        base = {"batch4/seed0": "e2e_8b_r3.pt", "batch4/seed1": "e2e_8b_r3b.pt"}.get(name)
        if tag in ("t1", "t2", "t4") and base and "hidden_m4" in c:
            h_self += _rel_l2(c["hidden_m4"], torch.load(os.path.join(golden_dir, base), weights_only=True)["cases"]["batch4/bf16"]["hidden_m4"])
    mean = lambda v: sum(v) / len(v)
    print(f"hidden[:, -4] relative L2 ({len(h_hip)} clips): hip vs ref bf16 {mean(h_hip):.4f}, the reference vs itself {mean(h_self):.4f} ({len(h_self)} pairs); hip vs ref fp32 "
          f"{mean(h32_hip):.4f}, the reference's bf16 pass vs its fp32 pass {mean(h32_ref):.4f}")
    assert len(h_hip) >= 32 and len(h_self) >= 24 and len(h32_hip) >= 32
    assert mean(h32_hip) <= REF_SELF_FACTOR * mean(h32_ref) and mean(h_hip) <= REF_SELF_FACTOR * max(mean(h_self), mean(h32_ref)), (mean(h_hip), mean(h_self), mean(h32_hip), mean(h32_ref))
    assert hip16[0] >= self16[0] - 0.01 and hip16[1] >= self16[1] - 0.01, (hip16, self16)
    assert hip32[0] >= self32[0] - 0.01 and hip32[1] >= self32[1] - 0.01, (hip32, self32)


def _pinned_cases(g, golden_dir):
    """(B, input seed, the reference's bf16 record) of the 13 recorded passes = 37 clips (e2e_8b_full.pt, e2e_8b_r3.pt / _r3b.pt, e2e_8b_r5.pt)."""
    cases = [(1, s, g["cases"][f"bf16/{s}"]) for s in sorted({int(k.split("/")[1]) for k in g["cases"] if k.startswith("bf16/")})]
    for f in ("e2e_8b_r3.pt", "e2e_8b_r3b.pt"):
        r = torch.load(os.path.join(golden_dir, f), weights_only=True)["cases"]["batch4/bf16"]
        cases.append((r["B"], r["seed"], r))
    c5 = torch.load(os.path.join(golden_dir, "e2e_8b_r5.pt"), weights_only=True)["cases"]
    return cases + [(4, seed, c5[f"batch4/seed{seed}/bf16"]) for seed in range(2, 8)]


def _level_flip_stats(got, r16, tie_ulps):
    """(flips, flips to a token outside r16's recorded top four, flips to a recorded token beyond ``tie_ulps`` of r16's winner) of ``got`` against r16."""
    flips = outside = beyond = 0
    for i in (got != r16["logit"]).nonzero().flatten().tolist():
        flips += 1
        ids, vals = r16["top_ids"][i].tolist(), r16["top_values"][i].tolist()
        if int(got[i]) not in ids:
            outside += 1
        elif (vals[0] - vals[ids.index(int(got[i]))]) / _bf16_ulp(vals[0]) > tie_ulps:
            beyond += 1
    return flips, outside, beyond


def _rel_l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm(dim=-1) / b.norm(dim=-1)).tolist()


def _corr(a, b):
    from scipy.stats import kendalltau, pearsonr, spearmanr
    return float(spearmanr(a, b)[0]), float(pearsonr(a, b)[0]), float(kendalltau(a, b)[0])


def _ref16_on_fp32_clips(cases):
    return [x for _B, _s, rec, _c, rec32 in cases if rec32 is not None for x in rec["score1"].float().tolist()]


def _ref_self_corr(golden_dir):
    """SRCC / PLCC / KRCC of the reference's bf16 scores under 1 / 2 / 4 host threads against its own 8-thread pass, pooled over the 13
    re-scored clips x 3 thread counts (tests/golden/e2e_8b_r5.pt): 0.9836 / 0.9975 / 0.9178."""
    ld = lambda f: torch.load(os.path.join(golden_dir, f), weights_only=True)["cases"]
    base = {f"one/{k.split('/')[1]}": c["score1"].float() for k, c in ld("e2e_8b_full.pt").items() if k.startswith("bf16/")}
    base["batch4/seed0"] = ld("e2e_8b_r3.pt")["batch4/bf16"]["score1"].float()
    base["batch4/seed1"] = ld("e2e_8b_r3b.pt")["batch4/bf16"]["score1"].float()
    x, y = [], []
    for key, c in ld("e2e_8b_r5.pt").items():
        name, tag = key.rsplit("/", 1)
        if tag in ("t1", "t2", "t4") and name in base:
            x += c["score1"].float().tolist(); y += base[name].tolist()
    assert len(x) == 39
    return _corr(x, y)


def test_full_size_8b_stage1_16_frames_matches_the_reference(full_8b, golden_r3):
    """The STAGE-1 flavour (internvl_chat_eval1/modeling_internvl_chat.py:250-366: the same pass, {'label', 'logit'} only, no score
    head) at full depth on a 16-frame clip (N = 4281: BASELINE config 4's clip shape at the 8B dims), against the reference's eval1
    class built around the same weights.  The model object is switched to stage 1 for the call: the stage decides which rows are
    consumed and what is returned, not the weights."""
    model, cfg, _g = full_8b
    r16, r32 = golden_r3["cases"]["stage1/bf16"], golden_r3["cases"]["stage1/fp32"]
    T, seed = r16["T"], r16["seed"]
    toks = synth.canonical_tokens(cfg, 1, T, seed=seed)
    model.img_context_token_id = toks["img_context_token_id"]
    dev = model.device
    model._native()            # the (stage-2) weights are on the device before the stage flips: nothing is re-uploaded under stage 1
    model.stage = 1
    try:
        out = model(mos=None, pixel_values=synth.synthetic_frames(T, 448, seed=seed).to(dev), input_ids=toks["input_ids"],
                    attention_mask=toks["attention_mask"], image_flags=torch.ones(T, 1, dtype=torch.long), labels=toks["labels"],
                    motion_feature=synth.synthetic_motion(1, cfg.motion_dim, seed=seed).to(dev))
        torch.cuda.synchronize()
    finally:
        model.stage = 2
    assert set(out) == {"label", "logit"}                                              # eval1's return dict (:365-366)
    rows = r16["answer_rows"]
    assert r16["n_rows"] == out["label"].numel() == toks["input_ids"].shape[1] - 1 == 4280
    assert torch.equal(out["label"].cpu()[rows], r16["label"])
    got = out["logit"].cpu()[rows]
    n_tie = _level_rows_ok(got, r16, "stage 1")
    agree_hip, agree_ref = int((got == r32["logit"]).sum()), int((r16["logit"] == r32["logit"]).sum())
    print(f"stage 1, 16 frames: level tokens {len(rows) - n_tie}/{len(rows)} identical to the bf16 reference, agreement with fp32: hip {agree_hip}, reference bf16 {agree_ref}")
    assert n_tie <= max(2, len(rows) // 4) and agree_hip >= agree_ref - 2


def test_full_size_8b_generate_matches_the_reference_cache_path(full_8b, golden_r3):
    """generate() (modeling_internvl_chat.py:769-811) at full depth: 12 greedy tokens behind clip 0's 8-frame prompt, against the
    REFERENCE's own KV-cache loop (modeling_internlm2.py:397-402, :1126-1163) on the golden's weights with five planted-margin
    lm-head rows (every step won by >= the recorded margin, so token equality is a hard assert), end-of-sequence checking on."""
    model, cfg, g = full_8b
    rec = golden_r3["cases"]["greedy/bf16"]
    toks = synth.canonical_tokens(cfg, 4, 8, seed=rec["seed"])
    n_prompt = int((toks["labels"][0] == -100).sum())
    assert n_prompt == rec["prompt_len"]
    ids = toks["input_ids"][:1, :n_prompt].clone()
    ctx = toks["img_context_token_id"]
    ids[0, (ids[0] == ctx).nonzero()[-1]] = 7                                          # as the generator: no motion slot in generate() prompts
    pv = synth.synthetic_frames(32, 448, seed=rec["seed"])[:8].to(model.device)
    model.img_context_token_id = ctx
    w = model.language_model.output.weight
    keep = w.data[rec["level_ids"]].clone()
    try:
        w.data[rec["level_ids"]] = keep * golden_r3["plant_scale"]
        model._invalidate()
        got = model.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=rec["n_new"], do_sample=False,
                             eos_token_id=toks["im_end_id"]).cpu()
        print("generate at full depth: hip", got.tolist(), "reference", rec["tokens"].tolist(), "margins (sigma)", [round(float(x), 2) for x in rec["margin_sigma"]])
        assert torch.equal(got, rec["tokens"])
    finally:
        w.data[rec["level_ids"]] = keep
        model._invalidate()


# ---------------------------------------------------------------------------------------------------------
# fp8 mode of the InternLM2 prefill linears (BASELINE config 5; aigv_set_precision).  The reference has no fp8 path:
# the HIP kernels are checked against oracle/fp8.py (the same definition evaluated on the CPU), and the distance to the
# bf16 result - the price of the mode - is measured and bounded.
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("trim", [True, False])
def test_fp8_llm_mode_matches_its_oracle(trim):
    from oracle import fp8 as O8
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=3)
    B, T, seed = 3, 2, 31
    model, sd, toks, pv, motion, ref, out = run_case(cfg, B=B, T=T, seed=seed)        # bf16 pass + bf16 oracle
    flags = torch.ones(B * T, 1, dtype=torch.long)
    with O8.fp8_llm(cfg.llm_config.num_hidden_layers):
        ref8 = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                              toks["img_context_token_id"], mos=None, stage=2, return_intermediates=True)
    kw = dict(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
              labels=toks["labels"], motion_feature=motion)
    model.set_precision("fp8")
    model.set_attention_numerics("reference")   # (the fp8 oracle rounds the score matrix like the reference)
    model.set_row_trimming(trim)
    try:
        out8 = model(**kw)
        all8 = model(full_logits=True, **kw)["logit"].cpu()                           # argmax of EVERY row (debug surface)
        torch.cuda.synchronize()
    finally:
        model.set_row_trimming(True)
        model.set_attention_numerics("fp32")
        model.set_precision("bf16")
    # 1. scores against the mode's own oracle: the bar of the bf16 path
    score_ok(out8["score1"], ref8["score1"])
    # 2. level tokens.  A one-ulp difference in a bf16 activation (fp32 summation order) can flip an e4m3 code downstream - a 6 % step on
    #    that element - so the HIP path sits further from its oracle than in bf16 mode, and these random-weight models have near-uniform
    #    vocabulary logits.  Bars: on the answer rows every disagreement is a near-tie of the ORACLE's own logits (<= 4 bf16 ulps) and at
    #    most a quarter of the rows; over all rows the HIP argmax agrees with the fp8 oracle far more often than the fp8 oracle agrees
    #    with the bf16 oracle (i.e. it implements THIS arithmetic, not merely something of similar accuracy).
    want = ref8["label"] != -100
    got, exp = out8["logit"].cpu()[want], ref8["logit"][want]
    logits = ref8["logits"][..., :-1, :].reshape(-1, ref8["logits"].shape[-1])[want]
    bad = (got != exp).nonzero().flatten().tolist()
    for r in bad:
        a_, b_ = logits[r, exp[r]].item(), logits[r, got[r]].item()
        ulp = 2.0 ** (torch.tensor(abs(a_)).clamp_min(1e-30).log2().floor().item() - 7)
        assert abs(a_ - b_) <= 4 * ulp, f"answer row {r}: argmax differs beyond a near-tie ({abs(a_ - b_) / ulp:.1f} ulp)"
    assert len(bad) <= max(1, int(want.sum()) // 4), len(bad)
    agree_own = float((all8 == ref8["logit"]).float().mean())
    agree_modes = float((ref8["logit"] == ref["logit"]).float().mean())
    print(f"all-row argmax agreement: hip fp8 vs fp8 oracle {agree_own:.3f}; fp8 oracle vs bf16 oracle {agree_modes:.3f}")
    assert agree_own >= 0.90 and agree_own >= agree_modes + 0.04
    # 3. the price of the mode: distance to the bf16 result (e4m3 keeps 3 mantissa bits per operand; the error averages over K)
    drift = (out8["score1"].float().cpu() - ref["score1"].float()).abs().max().item()
    drift_ref = (ref8["score1"].float() - ref["score1"].float()).abs().max().item()
    print(f"fp8 vs bf16 score drift: hip {drift:.4g}, oracle {drift_ref:.4g}")
    assert drift <= 0.05 and drift_ref <= 0.05
    # 4. switching back restores the bf16 result exactly
    again = model(**kw)
    assert torch.equal(again["score1"], out["score1"]) and torch.equal(again["logit"], out["logit"])


def test_fp8_mode_survives_a_weight_reload():
    """Reloading weights must requantise: the e4m3 copies of the old weights are dropped at finalize and the mode is re-applied."""
    cfg = pkg.tiny(image_size=224, vit_layers=1, llm_layers=2)
    B, T = 2, 2
    flags = torch.ones(B * T, 1, dtype=torch.long)
    res = {}
    model = None
    for order in ("fresh", "reloaded"):
        for seed in ((41,) if order == "fresh" else (40, 41)):
            sd = synth.make_state_dict(cfg, seed=seed, rich=True)
            if model is None or order == "fresh":
                model = make_model(cfg, sd)
            else:
                model.load_state_dict(sd)
            toks = synth.canonical_tokens(cfg, B, T, seed=41)
            model.img_context_token_id = toks["img_context_token_id"]
            kw = dict(mos=None, pixel_values=synth.synthetic_frames(B * T, 224, seed=41), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                      image_flags=flags, labels=toks["labels"], motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=41))
            model.set_precision("fp8")
            o8 = model(**kw)
            model.set_precision("bf16")
            o16 = model(**kw)
            res[(order, seed)] = (o8["score1"].clone(), o8["logit"].clone(), o16["score1"].clone(), o16["logit"].clone())
        if order == "fresh":
            model = None
    a, b = res[("fresh", 41)], res[("reloaded", 41)]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not torch.equal(a[1], a[3]) or not torch.equal(a[0], a[2])      # the two modes do differ on this input
