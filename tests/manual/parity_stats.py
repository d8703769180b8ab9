"""Full-size parity statistics on an MI355X: every clip the imported reference was recorded on (tests/golden/e2e_8b_full.pt: five
one-clip seeds; e2e_8b_r3.pt / e2e_8b_r3b.pt: two batches of four) scored under

    attention numerics {reference score rounding, fp32 scores}  x  GEMM dispatch modes {0 row plans, 1 128 kernel, 2 256 kernel}

Every (numerics, mode) pair is a CORRECT evaluation of the same arithmetic definition up to fp32 summation order (and, for the numerics,
up to where the score matrix rounds); the table shows how far each sits from the reference's recorded bf16 and fp32 scores, in bf16 ulps of
the score, next to the reference's own spread against itself (tests/golden/e2e_8b_r4_self.pt: host thread counts).

    python tests/manual/parity_stats.py [out.json]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def ulp(x):
    return 2.0 ** (torch.tensor(abs(float(x))).clamp_min(1e-30).log2().floor().item() - 7)


def main():
    g = torch.load(os.path.join(G, "e2e_8b_full.pt"), weights_only=True)
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=g["vision_config"], llm_config=g["llm_config"], force_image_size=448, select_layer=-1))
    dev = torch.device("cuda", 0)
    model = InternVLChatModel(cfg, device=dev, max_clips=4, max_frames=32, max_tokens=4 * synth.canonical_len(cfg, 8))
    sd = synth.make_state_dict(cfg, seed=g["w_seed"], rich=True)
    for k, v in g.get("overrides", {}).items():
        sd[k] = torch.full_like(sd[k], v)
    model.load_state_dict(sd)
    del sd
    model.eval()
    cases = []   # (name, B, seed, ref bf16 scores, ref fp32 scores, ref bf16 answer tokens, answer rows)
    for seed in (201, 202, 203, 204, 205):
        a, b = g["cases"][f"bf16/{seed}"], g["cases"][f"fp32/{seed}"]
        cases.append((f"one/{seed}", 1, seed, a["score1"].float(), b["score1"].float(), a["logit"], a["answer_rows"]))
    for f in ("e2e_8b_r3.pt", "e2e_8b_r3b.pt"):
        gg = torch.load(os.path.join(G, f), weights_only=True)["cases"]
        a, b = gg["batch4/bf16"], gg["batch4/fp32"]
        cases.append((f"batch4/seed{a['seed']}", 4, a["seed"], a["score1"].float(), b["score1"].float(), a["logit"], a["answer_rows"]))
    out = {}
    for numerics in ("reference", "fp32"):
        model.set_attention_numerics(numerics)
        for mode in (0, 1, 2):
            model.set_gemm_mode(mode)
            d16, d32, lev, nlev = [], [], 0, 0
            per_clip = []
            for name, B, seed, r16, r32, rlog, rows in cases:
                toks = synth.canonical_tokens(cfg, B, 8, seed=seed)
                model.img_context_token_id = toks["img_context_token_id"]
                o = model(mos=None, pixel_values=synth.synthetic_frames(B * 8, 448, seed=seed).to(dev), input_ids=toks["input_ids"],
                          attention_mask=toks["attention_mask"], image_flags=torch.ones(B * 8, 1, dtype=torch.long), labels=toks["labels"],
                          motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(dev))
                torch.cuda.synchronize()
                hip = o["score1"].float().cpu()
                for i in range(B):
                    u = ulp(r16[i])
                    d16.append(abs(float(hip[i] - r16[i])) / u)
                    d32.append(abs(float(hip[i] - r32[i])) / u)
                    per_clip.append(round(d16[-1], 1))
                got = o["logit"].cpu()[rows]
                lev += int((got != rlog).sum())
                nlev += int(rlog.numel())
            t = torch.tensor(d16)
            key = f"{numerics}/gemm{mode}"
            out[key] = dict(mean_ulps_vs_ref_bf16=float(t.mean()), max_ulps_vs_ref_bf16=float(t.max()), mean_ulps_vs_ref_fp32=float(torch.tensor(d32).mean()),
                            identical=int((t == 0).sum()),
                            level_tokens_differing=lev, level_rows=nlev, per_clip_ulps=per_clip)
            print(f"{key:18s} |hip - ref bf16| mean {t.mean():.2f} max {t.max():.1f} ulps, identical {int((t == 0).sum())}/13; vs ref fp32 mean "
                  f"{torch.tensor(d32).mean():.2f} ulps; level tokens differing {lev}/{nlev}; per clip {per_clip}", flush=True)
    # the reference's own two precisions, and its own spread against itself
    r = torch.tensor([abs(float(r16[i] - r32[i])) / ulp(r16[i]) for _n, B, _s, r16, r32, *_x in cases for i in range(B)])
    print(f"reference bf16 vs reference fp32: mean {r.mean():.2f} max {r.max():.1f} ulps")
    out["ref_bf16_vs_ref_fp32"] = dict(mean_ulps=float(r.mean()), max_ulps=float(r.max()))
    sp = os.path.join(G, "e2e_8b_r4_self.pt")
    if os.path.exists(sp):
        c = torch.load(sp, weights_only=True)["cases"]
        a, b = c["batch4/seed0/t8"]["score1"].float(), c["batch4/seed0/t4"]["score1"].float()
        s4 = [abs(float(a[i] - b[i])) / ulp(a[i]) for i in range(4)]
        s1 = abs(float(c["alone/seed0/clip0/t1"]["score1"].float()[0] - c["alone/seed0/clip0/t8"]["score1"].float()[0])) / ulp(a[0])
        print(f"reference vs itself: 8 vs 4 host threads {s4} ulps (mean {sum(s4) / 4:.2f}); 8 vs 1 threads, clip 0: {s1:.1f} ulps")
        out["reference_vs_itself"] = dict(threads_8_vs_4_ulps=s4, threads_8_vs_1_clip0_ulps=s1)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
