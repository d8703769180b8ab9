"""Price of the fp8 mode at full size: InternVL2-8B widths and depth, synthetic weights, 4 clips x 8 frames x 448 px per seed.
bf16 pass vs fp8 pass of the SAME model on the same inputs: score drift and answer-row argmax agreement.
python scripts/fp8_mode_drift.py [n_seeds]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = pkg.internvl2_8b(); T, B = 8, 4
N = synth.canonical_len(cfg, T); dev = torch.device("cuda", 0)
model = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * N).eval()
flags = torch.ones(B * T, 1, dtype=torch.long)
drifts, agree, scores = [], [], []
for seed in range(n_seeds):
    model.load_state_dict(synth.make_state_dict(cfg, seed=seed, device=dev, rich=True))
    toks = synth.canonical_tokens(cfg, B, T, seed=seed); model.img_context_token_id = toks["img_context_token_id"]
    pv = synth.synthetic_frames(B * T, cfg.image_size, seed=seed, device=dev)
    motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed, device=dev)
    kw = dict(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"], motion_feature=motion)
    model.set_precision("bf16"); a = model(**kw)
    model.set_precision("fp8"); b = model(**kw)
    torch.cuda.synchronize()
    want = (a["label"] != -100).cpu()
    d = (a["score1"].float() - b["score1"].float()).abs().cpu()
    ag = float((a["logit"].cpu()[want] == b["logit"].cpu()[want]).float().mean())
    drifts.append(d); agree.append(ag); scores.append(a["score1"].float().cpu())
    print(f"seed {seed}: score bf16 {[round(x, 4) for x in a['score1'].float().cpu().tolist()]} fp8 {[round(x, 4) for x in b['score1'].float().cpu().tolist()]} "
          f"max|d| {d.max().item():.4f}; answer-row argmax agreement {ag:.3f}", flush=True)
d = torch.cat(drifts)
print(f"over {len(d)} clips: score drift mean {d.mean().item():.4f} max {d.max().item():.4f} (bf16 ulp at 0.5 = 0.0039); "
      f"answer-row argmax agreement {sum(agree) / len(agree):.3f} (random-weight models: near-uniform vocabulary logits)")
