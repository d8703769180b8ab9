"""Randomised differential run of eval_utils.batched against the plain one-clip-per-call loop (tiny model, one MI355X): item streams that mix frame counts (2 to 16
frames per clip), decoded-frame sizes, ragged prompt lengths, uint8 frames and normalised pixel_values, items with and without `mos`; loop settings k = 1..6, look-ahead on /
off, graph replay on / off, model contexts smaller than k (the context grows), stage 1 and stage 2, the native SlowFast branch or a precomputed motion feature.  Every yielded
(score1, logit, label, loss) must equal the plain loop's bit for bit, in the stream's order.

    python tests/manual/fuzz_batched.py [n_rounds = 40] [seed = 0] [--8b]        # --8b: InternVL2-8B sizes, two long-lived models (use ~8 rounds)"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import eval_utils, synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402
from aigv_assessor_amd.slowfast import SlowFastR50  # noqa: E402

BIG = "--8b" in sys.argv          # InternVL2-8B sizes: the 256-tile GEMM row plans, split-K tails and the 448 px ingest instead of the tiny model's generic forms; the two
sys.argv = [a for a in sys.argv if a != "--8b"]      # models (plain / under test) are built once and live through all rounds
n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = pkg.internvl2_8b() if BIG else pkg.tiny(image_size=224, vit_layers=2, llm_layers=2)
sd = synth.make_state_dict(cfg, seed=5, rich=True, **({"device": torch.device("cuda", 0)} if BIG else {}))
sf_sd = synth.slowfast_state_dict(seed=3)
_CACHE = {}
print(f"model: InternViT {cfg.vision_config.hidden_size} x {cfg.vision_config.num_hidden_layers}, LLM {cfg.llm_config.hidden_size} x {cfg.llm_config.num_hidden_layers}, {cfg.image_size} px", flush=True)


def make_model(stage, max_clips, branch, role="dut"):
    if BIG:                                   # one long-lived model per (role, stage); the motion branch is switched per round
        m = _CACHE.get((role, stage))
        if m is None:
            m = _CACHE[(role, stage)] = InternVLChatModel(cfg, stage=stage, max_clips=1, device=torch.device("cuda", 0))
            m.load_state_dict(sd)
            m.eval()
            m._branch = SlowFastR50(sf_sd)
        m.slowfast_model = m._branch if branch else None
        return m
    m = InternVLChatModel(cfg, stage=stage, max_clips=max_clips)
    m.load_state_dict(sd)
    m.eval().cuda()
    if branch:
        m.slowfast_model = SlowFastR50(sf_sd)
    return m


def make_items(rng, n, branch, g):
    items, ctx = [], None
    T, hw = rng.choice([8, 12] if branch else [4, 8]), rng.choice([(300, 400), (240, 320)])
    for i in range(n):
        if rng.random() < 0.25:                                     # a change of frame geometry in mid-stream: the group in flight ends there
            T, hw = rng.choice([8, 12, 16] if branch else [2, 4, 8]), rng.choice([(300, 400), (240, 320), (224, 224)])   # (the native SlowFast branch takes T = 8..32, a multiple of 4)
        toks = synth.canonical_tokens(cfg, 1, T, seed=1000 + i)
        ctx = toks["img_context_token_id"]
        ids, lab = toks["input_ids"], toks["labels"]
        extra = rng.choice([0, 0, 1, 2, 5])
        if extra:
            a0 = int((lab[0] != -100).nonzero()[0])
            ids = torch.cat([ids[:, :a0], torch.randint(3, cfg.llm_config.vocab_size - 16, (1, extra), generator=g), ids[:, a0:]], 1)
            lab = torch.cat([lab[:, :a0], torch.full((1, extra), -100), lab[:, a0:]], 1)
        it = {"input_ids": ids, "labels": lab, "attention_mask": torch.ones_like(ids, dtype=torch.bool),
              "image_flags": torch.ones(1, T, 1, dtype=torch.long) if rng.random() < 0.5 else torch.ones(1, T, dtype=torch.long),
              "frames": torch.randint(0, 256, (T,) + hw + (3,), dtype=torch.uint8, generator=g).pin_memory()}
        if rng.random() < 0.7:
            it["mos"] = torch.tensor([rng.random()])
        if not branch:
            it["motion_feature"] = synth.synthetic_motion(1, cfg.motion_dim, seed=i).cuda()
        items.append(it)
    return items, ctx


def plain(model, items, stage):
    rows = []
    for it in items:
        out = model(mos=it["mos"][0].to(torch.bfloat16) if "mos" in it else None, pixel_values=model.ingest_frames(it["frames"].cuda()), input_ids=it["input_ids"],
                    attention_mask=it["attention_mask"], image_flags=it["image_flags"][0].reshape(-1, 1), labels=it["labels"],
                    **({"motion_feature": it["motion_feature"]} if "motion_feature" in it else {}))
        rows.append({k: (v.cpu().clone() if torch.is_tensor(v) else v) for k, v in out.items()})
    return rows


bad = 0
for r in range(n_rounds):
    rng = random.Random(seed0 * 1000 + r)
    g = torch.Generator().manual_seed(seed0 * 1000 + r)
    stage = rng.choice([2, 2, 2, 1])
    branch = rng.random() < 0.6
    k = rng.randint(1, 6)
    max_clips = rng.choice([1, 2, 4, 6])
    ahead, graph, as_pv = rng.random() < 0.7, rng.random() < 0.6, rng.random() < 0.25
    n = rng.randint(1, 8 if BIG else 14)
    items, ctx = make_items(rng, n, branch, g)
    ref_model = make_model(stage, 1, branch, role="ref")
    ref_model.enable_graph_replay(False)
    ref_model.img_context_token_id = ctx
    want = plain(ref_model, items, stage)
    model = make_model(stage, max_clips, branch)
    model.img_context_token_id = ctx
    model.enable_graph_replay(graph)
    if as_pv:                                                       # the dataloader's own form: normalised pixel_values [1, T, 3, S, S]
        for it in items:
            it["pixel_values"] = ref_model.ingest_frames(it["frames"].cuda()).float().cpu()[None]
    passes = 2 if graph else 1                                      # (a second pass over the same stream replays what the first one captured)
    for p in range(passes):
        got = list(eval_utils.batched(items, model, k=k, ahead=ahead, frames=None if as_pv else (lambda it: it["frames"])))
        ok = len(got) == len(items) and all(a is b for (a, _), b in zip(got, items))
        for (it, out), w in zip(got, want):
            ok = ok and torch.equal(out["logit"], w["logit"]) and torch.equal(out["label"], w["label"]) and out["logit"].shape == (it["input_ids"].shape[1] - 1,)
            if stage == 2:
                ok = ok and torch.equal(out["score1"], w["score1"])
                ok = ok and ((out["loss"] is None) == (w.get("loss") is None)) and (out["loss"] is None or torch.equal(out["loss"], w["loss"]))
            else:
                ok = ok and "score1" not in out
        if not ok:
            bad += 1
            print(f"MISMATCH round {r} pass {p}: stage {stage} branch {branch} k {k} max_clips {max_clips} ahead {ahead} graph {graph} as_pv {as_pv} n {n}", flush=True)
    del model, ref_model
    if r % 10 == 9:
        print(f"round {r + 1}/{n_rounds}: mismatching rounds so far {bad}", flush=True)
assert bad == 0, bad
print(f"FUZZ_OK {n_rounds} rounds")
