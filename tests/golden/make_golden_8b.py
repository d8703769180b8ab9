"""Full-size golden vectors: the REFERENCE itself (imported, CPU eager path) at InternVL2-8B size — InternViT-300M with 24
layers + InternLM2.5-7B with 32 layers, one 8-frame 448x448 clip, N = 2177 — the configuration BASELINE.json's metric is
quoted on (reference: internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:306-488, stage2_eval.py:930-941).

Run (build container only; needs ~50 GB of RAM and ~30 min on 8 cores):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_8b.py

What is recorded (outputs only; the weights are regenerated from the seed by ``synth.make_state_dict`` on the CPU generator):

* ONE seeded weight set (seed W_SEED, with the OVERRIDES below; the reference's config.json dims incl. its dynamic-NTK rope_scaling entry), run on
  the inputs of several input seeds, in fp32 AND bf16: ``score1``, the answer-row argmax ids, the reference's top-4 logit
  values / ids on every answer row (for the tie rule of the level check), a subsample of ``hidden[:, -4]``.
* the same weights with a PLANTED MARGIN: five "quality level" rows of ``language_model.output.weight`` scaled by 8 (exact
  in bf16).  The five ids are chosen by a seeded search over the reference's own answer-row logits so that on every answer
  row one of them wins by a wide margin (>= MARGIN_SIGMA standard deviations of that row's vocabulary logits, i.e. far
  outside bf16 accumulation noise) and at least three different ids win somewhere.  With such weights "quality levels
  bit-exact" is a hard assert with no tie exemption.

Output: tests/golden/e2e_8b_full.pt  (a few hundred KB)
"""
import contextlib
import io
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402

W_SEED = 101
INPUT_SEEDS = (201, 202, 203, 204, 205)
PLANT_INPUT_SEED = 201
PLANT_SCALE = 8.0
MARGIN_SIGMA = 0.75          # top-1 minus top-2 of the planted logits, in units of the row's (unscaled) vocabulary-logit sigma x scale
B, T = 1, 8
# synth's score head is calibrated for narrow test models; at 4096 wide and full depth its output sits around 0.1 +- 0.3 and the
# final ReLU clamps half of the clips to exactly 0.  The last bias is therefore set to 1.0 (exact in bf16), which keeps score1 of
# these inputs inside (0, 1.3) - a clamped score would compare equal whatever the arithmetic.  The test applies the same override.
OVERRIDES = {"mlpscore.fc5.bias": 1.0}


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def reference_dims():
    """llm_config / vision_config of the reference's shipped config.json (the only model config in its tree)."""
    with open(os.path.join(ref_shims.REF_ROOT, "internvl/model/internvl_chat_eval2/config.json")) as f:
        c = json.load(f)
    keep_l = ("architectures", "hidden_size", "intermediate_size", "num_attention_heads", "num_key_value_heads", "num_hidden_layers",
              "vocab_size", "rms_norm_eps", "rope_theta", "max_position_embeddings", "rope_scaling", "bias", "hidden_act", "pad_token_id")
    keep_v = ("architectures", "hidden_size", "intermediate_size", "num_attention_heads", "num_hidden_layers", "image_size", "patch_size",
              "layer_norm_eps", "norm_type", "qkv_bias", "qk_normalization", "hidden_act", "initializer_factor")
    llm = {k: c["llm_config"][k] for k in keep_l}
    vis = {k: c["vision_config"][k] for k in keep_v}
    llm["attn_implementation"] = "eager"          # no flash_attn here: the reference's CPU eager path, the one the metric names
    vis["use_flash_attn"] = False
    vis["drop_path_rate"] = 0.0                   # eval mode: DropPath is the identity either way
    return llm, vis


def inputs(cfg, seed, dt):
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    pv = synth.synthetic_frames(B * T, 448, seed=seed, dtype=dt)
    motion = synth.synthetic_motion(B, 2304, seed=seed, dtype=dt)
    return toks, pv, motion


def run(model, SlowFastStandIn, cfg, seed, dt):
    toks, pv, motion = inputs(cfg, seed, dt)
    model.img_context_token_id = toks["img_context_token_id"]
    SlowFastStandIn.feature = motion
    grabbed = {}
    hooks = [model.language_model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.logits)),
             model.language_model.model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("hidden", o.last_hidden_state))]
    t0 = time.time()
    with torch.no_grad(), quiet():
        out = model(mos=torch.full((B,), 0.5, dtype=dt), pixel_values=pv, input_ids=toks["input_ids"],
                    attention_mask=toks["attention_mask"], image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=toks["labels"])
    for h in hooks:
        h.remove()
    want = out["label"] != -100
    rows = grabbed["logits"][..., :-1, :].reshape(-1, grabbed["logits"].shape[-1])[want].float()      # [answer rows, V]
    top_v, top_i = rows.topk(4, dim=-1)
    rec = dict(seed=seed, dtype=str(dt), score1=out["score1"].clone(), label=out["label"][want].clone(), logit=out["logit"][want].clone(),
               n_rows=int(out["label"].numel()), answer_rows=want.nonzero().flatten().clone(), top_values=top_v.clone(), top_ids=top_i.clone(),
               row_sigma=rows.std(dim=-1).clone(), hidden_m4_sub=grabbed["hidden"][:, -4, ::16].clone(), seconds=time.time() - t0)
    print(f"  seed {seed} {dt}: score1 {out['score1'].float().tolist()} answer argmax {out['logit'][want].tolist()} ({rec['seconds']:.0f} s)", flush=True)
    return rec, rows


def choose_level_ids(rows, hi):
    """Five token ids such that, with their output rows scaled by PLANT_SCALE, every answer row is won by one of them with a wide
    margin.  Seeded random search over the reference's own logits; candidates are drawn from the ids that score high on some row
    (the hidden states of a random-weight model share a dominant direction, so most ids never win anything).  Returns the ids and
    the margin (in sigmas) that was reached: the requested MARGIN_SIGMA with >= 3 distinct winners if the data allow it, else the
    best relaxation that does."""
    g = torch.Generator().manual_seed(4242)
    sigma = rows.std(dim=-1, keepdim=True)
    z = rows[:, :hi] / sigma                                   # standardised logits [R, hi]
    other_max = rows.max(dim=-1).values / sigma[:, 0]          # the best unscaled competitor, in sigmas
    pool = z.max(dim=0).values.topk(4000).indices              # ids that are strong on at least one row
    pool = pool[pool >= 3]
    for margin, need_distinct in ((MARGIN_SIGMA, 3), (MARGIN_SIGMA, 2), (0.5, 3), (0.5, 2), (0.5, 1), (0.3, 1)):
        for attempt in range(40):
            cand = pool[torch.randint(0, pool.numel(), (200000, 5), generator=g)]
            zc = z[:, cand] * PLANT_SCALE                      # [R, n, 5]
            top2 = zc.topk(2, dim=-1).values
            gap = (top2[..., 0] - top2[..., 1]) / PLANT_SCALE  # in sigmas of the unscaled logits
            clear = top2[..., 0] - other_max[:, None]          # above every unscaled logit
            ok = (gap >= margin).all(0) & (clear >= 2.0).all(0)
            distinct = torch.zeros(cand.shape[0], dtype=torch.long)
            win = zc.argmax(-1)                                # [R, n]
            for k in range(5):
                distinct += (win == k).any(0).long()
            ok &= distinct >= need_distinct
            ok &= (cand.sort(-1).values.diff(dim=-1) != 0).all(-1)
            if bool(ok.any()):
                j = int(ok.nonzero()[0])
                best = cand[j].tolist()
                print(f"  planted level ids {best} (margin >= {margin} sigma, {need_distinct}+ distinct winners); min gap "
                      f"{gap[:, j].min().item():.2f} sigma, winners {win[:, j].tolist()}", flush=True)
                return best, margin
    raise RuntimeError("no level-id set with a usable margin")


def main():
    llm, vis = reference_dims()
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=vis, llm_config=llm, force_image_size=448, select_layer=-1))
    m2, _m1, cfg2, SlowFastStandIn = ref_shims.install(llm, vis)
    t0 = time.time()
    with quiet():
        rcfg = cfg2.InternVLChatConfig(select_layer=-1, force_image_size=448, downsample_ratio=0.5, template="internlm2-chat", ps_version="v2")
        model = m2.InternVLChatModel(rcfg).eval()              # fp32 parameters
    print(f"reference model constructed in {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    sd = synth.make_state_dict(cfg, seed=W_SEED, rich=True)    # bf16 values; the fp32 model takes their exact upcast
    for k, v in OVERRIDES.items():
        sd[k] = torch.full_like(sd[k], v)
    model.load_state_dict(sd, strict=True)
    del sd
    print(f"seeded weights generated and loaded in {time.time() - t0:.0f} s", flush=True)

    out = dict(llm_config=llm, vision_config=vis, w_seed=W_SEED, B=B, T=T, plant_scale=PLANT_SCALE, overrides=OVERRIDES, cases={})
    print("fp32 passes", flush=True)
    for s in INPUT_SEEDS:
        out["cases"][f"fp32/{s}"], _ = run(model, SlowFastStandIn, cfg, s, torch.float32)
    model = model.to(torch.bfloat16)                           # exact: every value is a bf16 number already
    print("bf16 passes", flush=True)
    plant_rows = None
    for s in INPUT_SEEDS:
        out["cases"][f"bf16/{s}"], rows = run(model, SlowFastStandIn, cfg, s, torch.bfloat16)
        if s == PLANT_INPUT_SEED:
            plant_rows = rows
    ids, margin = choose_level_ids(plant_rows, hi=92000)
    with torch.no_grad():
        model.language_model.output.weight[ids] *= PLANT_SCALE
    print("planted-margin pass (bf16)", flush=True)
    rec, rows = run(model, SlowFastStandIn, cfg, PLANT_INPUT_SEED, torch.bfloat16)
    rec["level_ids"] = ids
    top2 = rows.topk(2, dim=-1).values
    rec["margin_sigma"] = ((top2[:, 0] - top2[:, 1]) / (plant_rows.std(dim=-1) * PLANT_SCALE)).clone()
    assert all(int(t) in ids for t in rec["logit"].tolist()), "a non-level token won a planted row"
    assert float(rec["margin_sigma"].min()) >= margin * 0.99
    rec["margin_floor"] = margin
    out["cases"][f"planted/{PLANT_INPUT_SEED}"] = rec
    torch.save(out, os.path.join(HERE, "e2e_8b_full.pt"))
    print("wrote", os.path.join(HERE, "e2e_8b_full.pt"), flush=True)


if __name__ == "__main__":
    main()
