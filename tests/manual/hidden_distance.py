"""How far is one evaluation of the scorer from another, measured on the 4096-wide hidden state the score head reads (hidden[:, -4], after the
final RMSNorm) instead of on the scalar score?  Relative L2 distance, per clip, of

    HIP (both attention numerics)          vs the reference's bf16 pass (8 host threads)
    the reference under 1 / 2 / 4 threads  vs its 8-thread pass
    the reference's bf16 pass              vs its fp32 pass;      HIP vs its fp32 pass

on the iid weight set (tests/golden/e2e_8b_r3.pt / _r3b.pt / _r5.pt: 32 clips in batches of four) and on the CONDITIONED set
(e2e_8b_conditioned.pt).  A vector norm averages over 4096 coordinates, so a few clips already separate "the same noise level" from "twice the noise".

    python tests/manual/hidden_distance.py        # MI355X"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm(dim=-1) / b.norm(dim=-1)).tolist()


def st(v):
    return f"mean {sum(v) / len(v):.4f} max {max(v):.4f} (n = {len(v)})" if v else "-"


def run(tag, cfg, sd, batches):
    """batches: {seed: {"t8": rec, "t4": rec?, "fp32": rec?, ...}}"""
    dev = torch.device("cuda", 0)
    model = InternVLChatModel(cfg, device=dev, max_clips=4, max_frames=32, max_tokens=4 * synth.canonical_len(cfg, 8))
    model.load_state_dict(sd)
    model.eval()
    self_d, ref32 = [], []
    for s, b in batches.items():
        for t in ("t1", "t2", "t4"):
            if t in b:
                self_d += rel(b[t]["hidden_m4"], b["t8"]["hidden_m4"])
        if "fp32" in b:
            ref32 += rel(b["t8"]["hidden_m4"], b["fp32"]["hidden_m4"])
    print(f"[{tag}] reference vs itself (other thread counts vs 8): {st(self_d)};  reference bf16 vs its fp32 pass: {st(ref32)}")
    for numerics in ("fp32", "reference"):
        model.set_attention_numerics(numerics)
        d16, d32 = [], []
        for s, b in batches.items():
            toks = synth.canonical_tokens(cfg, 4, 8, seed=s)
            model.img_context_token_id = toks["img_context_token_id"]
            model(mos=None, pixel_values=synth.synthetic_frames(32, 448, seed=s).to(dev), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                  image_flags=torch.ones(32, 1, dtype=torch.long), labels=toks["labels"], motion_feature=synth.synthetic_motion(4, cfg.motion_dim, seed=s).to(dev))
            h = model.last_hidden_rows(4).cpu()
            d16 += rel(h, b["t8"]["hidden_m4"])
            if "fp32" in b:
                d32 += rel(h, b["fp32"]["hidden_m4"])
        print(f"[{tag}] hip, attention numerics {numerics:9s}: vs reference bf16 {st(d16)};  vs reference fp32 {st(d32)}")
    del model
    torch.cuda.empty_cache()


g = torch.load(os.path.join(G, "e2e_8b_full.pt"), weights_only=True)
cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=g["vision_config"], llm_config=g["llm_config"], force_image_size=448, select_layer=-1))
sd = synth.make_state_dict(cfg, seed=g["w_seed"], rich=True)
for k, v in g.get("overrides", {}).items():
    sd[k] = torch.full_like(sd[k], v)
c5 = torch.load(os.path.join(G, "e2e_8b_r5.pt"), weights_only=True)["cases"]
iid = {}
for f in ("e2e_8b_r3.pt", "e2e_8b_r3b.pt"):
    c = torch.load(os.path.join(G, f), weights_only=True)["cases"]
    s = c["batch4/bf16"]["seed"]
    iid[s] = {"t8": c["batch4/bf16"], "fp32": c["batch4/fp32"]}
    for t in ("t1", "t2", "t4"):
        if f"batch4/seed{s}/{t}" in c5:
            iid[s][t] = c5[f"batch4/seed{s}/{t}"]
for s in range(2, 8):
    iid[s] = {"t8": c5[f"batch4/seed{s}/bf16"]}
    if f"batch4/seed{s}/fp32" in c5:
        iid[s]["fp32"] = c5[f"batch4/seed{s}/fp32"]
run("iid weights", cfg, sd, iid)
cond_path = os.path.join(G, "e2e_8b_conditioned.pt")
if os.path.exists(cond_path):
    cc = torch.load(cond_path, weights_only=True)["cases"]
    synth.condition_state_dict(sd, cfg)
    cond = {}
    for k, v in cc.items():
        _b, seed, prec, t = k.split("/")
        cond.setdefault(int(seed[4:]), {})["fp32" if prec == "fp32" else t] = v
    run("conditioned weights", cfg, sd, cond)
