"""Follow-up of scripts/store_policy_ab.py: WHICH part of the step changes under gemm256_variant = 5 (diagnostic build)?  Per kernel class times
(prof_enable) and where the first differing intermediate appears (ViT tokens, projected tokens, scores)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AIGV_AMD_LIB", os.path.join(ROOT, "scripts", "_abl", "libaigv_store_ab.so"))
sys.path.insert(0, ROOT)
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel

cfg = pkg.internvl2_8b()
B, T = 4, 8
dev = torch.device("cuda", 0)
N = synth.canonical_len(cfg, T)
model = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * N)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
toks = synth.canonical_tokens(cfg, B, T, seed=0)
model.img_context_token_id = toks["img_context_token_id"]
model.eval()
pv = synth.synthetic_frames(B * T, 448, seed=0).to(dev)
motion = synth.synthetic_motion(B, cfg.motion_dim, seed=0).to(dev)
flags = torch.ones(B * T, 1, dtype=torch.long)
step = lambda: model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"], motion_feature=motion)
base = {}
for v in (0, 5, 0, 5, 6):
    model.tune("gemm256_variant", v)
    for _ in range(2):
        o = step()
    torch.cuda.synchronize()
    vt = model.vit_tokens(pv)
    pj = model.project(vt)
    torch.cuda.synchronize()
    model.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        o = step()
    e1.record()
    torch.cuda.synchronize()
    p = model.prof_read()
    model.prof_enable(False)
    cur = dict(vit=vt.clone(), proj=pj.clone(), score=o["score1"].clone(), logit=o["logit"].clone())
    if v == 0 and not base:
        base = cur
    print(f"variant {v}: {e0.elapsed_time(e1) / 5:7.2f} ms/step (with events);  " + "  ".join(f"{k} {x['ms'] / 5:6.2f} ms / {x['launches'] // 5} launches" for k, x in p.items() if x["launches"])
          + ";  differing vs the first arm: " + ", ".join(f"{k} {int((cur[k] != base[k]).sum())}/{cur[k].numel()}" for k in cur), flush=True)
