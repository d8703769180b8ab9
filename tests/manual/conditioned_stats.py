"""HIP against the reference on the CONDITIONED weight set (tests/golden/e2e_8b_conditioned.pt), by hand: every recorded batch under both
attention numerics, distances to the reference's bf16 (8-thread) and fp32 scores in bf16 ulps, next to the reference against itself (other thread counts).

    python tests/manual/conditioned_stats.py          # MI355X; ~1 min of CPU weight generation"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402


def ulp(x):
    return 2.0 ** (torch.tensor(abs(float(x))).clamp_min(1e-30).log2().floor().item() - 7)


def st(v):
    return f"mean {sum(v) / len(v):.2f} max {max(v):.1f} (n = {len(v)}) {[round(x, 1) for x in v]}" if v else "-"


c = torch.load(os.path.join(ROOT, "tests", "golden", "e2e_8b_conditioned.pt"), weights_only=True)
cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=c["vision_config"], llm_config=c["llm_config"], force_image_size=448, select_layer=-1))
dev = torch.device("cuda", 0)
model = InternVLChatModel(cfg, device=dev, max_clips=4, max_frames=32, max_tokens=4 * synth.canonical_len(cfg, 8))
sd = synth.condition_state_dict(synth.make_state_dict(cfg, seed=c["w_seed"], rich=True), cfg)
for k, v in c["overrides"].items():
    sd[k] = torch.full_like(sd[k], v)
model.load_state_dict(sd)
del sd
model.eval()
cases = c["cases"]
seeds = sorted({int(k.split("/")[1][4:]) for k in cases if k.endswith("/bf16/t8")})
self_d, ref32 = [], []
for s in seeds:
    a = cases[f"batch4/seed{s}/bf16/t8"]["score1"].float()
    for t in (1, 2, 4):
        o = cases.get(f"batch4/seed{s}/bf16/t{t}")
        if o is not None:
            self_d += [abs(float(o["score1"].float()[i] - a[i])) / ulp(a[i]) for i in range(4)]
    f = cases.get(f"batch4/seed{s}/fp32/t8")
    if f is not None:
        ref32 += [abs(float(a[i] - f["score1"].float()[i])) / ulp(a[i]) for i in range(4)]
print("reference vs itself (other host thread counts vs 8):", st(self_d))
print("reference bf16 vs its fp32 pass:                    ", st(ref32))
def corr(a, b):
    from scipy.stats import pearsonr, spearmanr
    return f"SRCC {spearmanr(a, b)[0]:.4f} PLCC {pearsonr(a, b)[0]:.4f}"


for precision in ("bf16", "fp8"):
    model.set_precision(precision)
    for numerics in ("fp32", "reference"):
        model.set_attention_numerics(numerics)
        d16, d32, lev, rows, s_hip, s_ref, s_hip32, s_ref32 = [], [], 0, 0, [], [], [], []
        for s in seeds:
            r8 = cases[f"batch4/seed{s}/bf16/t8"]
            toks = synth.canonical_tokens(cfg, 4, 8, seed=s)
            model.img_context_token_id = toks["img_context_token_id"]
            o = model(mos=None, pixel_values=synth.synthetic_frames(32, 448, seed=s).to(dev), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"],
                      image_flags=torch.ones(32, 1, dtype=torch.long), labels=toks["labels"], motion_feature=synth.synthetic_motion(4, cfg.motion_dim, seed=s).to(dev))
            hip, a = o["score1"].float().cpu(), r8["score1"].float()
            s_hip += hip.tolist(); s_ref += a.tolist()
            d16 += [abs(float(hip[i] - a[i])) / ulp(a[i]) for i in range(4)]
            f = cases.get(f"batch4/seed{s}/fp32/t8")
            if f is not None:
                d32 += [abs(float(hip[i] - f["score1"].float()[i])) / ulp(a[i]) for i in range(4)]
                s_hip32 += hip.tolist(); s_ref32 += f["score1"].float().tolist()
            lev += int((o["logit"].cpu()[r8["answer_rows"]] != r8["logit"]).sum()); rows += int(r8["logit"].numel())
        print(f"hip {precision}, attention numerics {numerics:9s}: vs ref bf16 {st(d16)};  vs ref fp32 {st(d32)};  level tokens differing {lev}/{rows};  task level vs ref bf16: {corr(s_hip, s_ref)}, vs ref fp32: {corr(s_hip32, s_ref32)}")
# the reference against itself at task level (every recorded thread count against the 8-thread pass, pooled)
x, y = [], []
for s in seeds:
    a = cases[f"batch4/seed{s}/bf16/t8"]["score1"].float().tolist()
    for t in (1, 2, 4):
        o = cases.get(f"batch4/seed{s}/bf16/t{t}")
        if o is not None:
            x += o["score1"].float().tolist(); y += a
print(f"reference vs itself, task level ({len(x)} pairs pooled): {corr(x, y)}")
x, y = [], []
for s in seeds:
    f = cases.get(f"batch4/seed{s}/fp32/t8")
    if f is not None:
        x += cases[f"batch4/seed{s}/bf16/t8"]["score1"].float().tolist(); y += f["score1"].float().tolist()
print(f"reference bf16 (8 threads) vs its fp32 pass, task level ({len(x)} clips): {corr(x, y)}")
