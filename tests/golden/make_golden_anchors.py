"""What tests/test_gpu_e2e.py used to compute with the CPU oracle on the GPU box for the two 4096-wide shallow fixtures (e2e.pt,
e2e_llama.pt), recorded once from the REFERENCE itself in this container instead (round 6: the live oracle passes cost ~40 s of the
driver's 900 s `-m gpu` limit on a slow host):

  * the reference's fp32 pass over the bf16-ROUNDED weights and inputs of every bf16 case - the anchor of score_near_fp32
    (the fixtures' own fp32_b1 case runs un-rounded fp32 weights: another quantity);
  * the reference's bf16 logits of the answer rows, top 8 per row (ids + values) - what the level-token tie rule reads.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_anchors.py

Weights are not stored (seeded: synth.make_state_dict).  Output: tests/golden/e2e_anchors.pt
"""
import contextlib
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth, weights  # noqa: E402
import make_golden as G  # noqa: E402
import make_golden_llama as GL  # noqa: E402


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def run(family, llm, vis, cases, out):
    m2, _m1, cfg2, SlowFastStandIn = ref_shims.install(llm, vis)
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=vis, llm_config=llm, force_image_size=448, select_layer=-1))
    for B, T, seed in cases:
        sd16 = synth.make_state_dict(cfg, seed=seed, dtype=torch.bfloat16, rich=True)
        if family == "llama":
            sd16 = weights.internlm2_to_llama(sd16, cfg.llm_config)
        toks = synth.canonical_tokens(cfg, B, T, seed=seed)
        rec = {"seed": seed, "B": B, "T": T}
        for dt in (torch.bfloat16, torch.float32):
            with quiet():
                rcfg = cfg2.InternVLChatConfig(select_layer=-1, force_image_size=448, downsample_ratio=0.5, template="internlm2-chat", ps_version="v2")
                model = m2.InternVLChatModel(rcfg).to(dt).eval()
            if family == "llama":   # (as make_golden_llama.py: the rotary inv_freq buffer stays fp32 in a checkpoint loaded the reference's way)
                rot = model.language_model.model.rotary_emb
                inv, _ = rot.compute_default_rope_parameters(rot.config)
                rot.inv_freq = inv
                rot.original_inv_freq = inv.clone()
            model.load_state_dict({k: v.to(dt) for k, v in sd16.items()}, strict=True)       # bf16-rounded values in both passes
            model.img_context_token_id = toks["img_context_token_id"]
            pv = synth.synthetic_frames(B * T, 448, seed=seed).to(dt)
            SlowFastStandIn.feature = synth.synthetic_motion(B, 2304, seed=seed).to(dt)
            grabbed = {}
            h = model.language_model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.logits))
            with torch.no_grad(), quiet():
                o = model(mos=torch.full((B,), 0.5, dtype=dt), pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                          image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=toks["labels"])
            h.remove()
            if dt == torch.float32:
                rec["score1_fp32_of_bf16_weights"] = o["score1"].clone()
            else:
                want = o["label"] != -100
                lg = grabbed["logits"][..., :-1, :].reshape(-1, grabbed["logits"].shape[-1])[want].float()
                top = lg.topk(8, dim=-1)
                rec.update(score1_bf16=o["score1"].clone(), answer_logit=o["logit"][want].clone(), top_ids=top.indices.clone(), top_values=top.values.clone())
            del model
        out[f"{family}/{seed}"] = rec
        print(family, seed, "fp32-of-bf16-weights score", rec["score1_fp32_of_bf16_weights"].tolist(), "bf16 score", rec["score1_bf16"].float().tolist())


def main():
    out = {"torch": str(torch.__version__)}
    run("internlm2", G.E2E_LLM, G.E2E_VIS, [(1, 8, 21), (2, 8, 22)], out)
    run("llama", GL.LLM, GL.VIS, [(1, 8, 31), (2, 8, 32)], out)
    # the recorded bf16 pass must BE the one the older fixtures hold (same reference, same container class)
    for fam, f, seeds in (("internlm2", "e2e.pt", {21: "bf16_b1", 22: "bf16_b2"}), ("llama", "e2e_llama.pt", {31: "bf16_b1", 32: "bf16_b2"})):
        g = torch.load(os.path.join(HERE, f), weights_only=True)
        for seed, tag in seeds.items():
            r, c = out[f"{fam}/{seed}"], g[tag]
            assert torch.equal(r["score1_bf16"], c["score1"]) and torch.equal(r["answer_logit"], c["logit"][c["label"] != -100]), (fam, seed)
    torch.save(out, os.path.join(HERE, "e2e_anchors.pt"))
    print("wrote e2e_anchors.pt")


if __name__ == "__main__":
    main()
