"""Four quality perspectives per clip on the headline config: the reference runs one full pass per perspective;
this path can share the ViT tokens, and further the whole video prefix of the LLM pass (forward_shared_prefix).
    python scripts/perspectives_bench.py [--clips 4] [--frames 8] [--prompts 4] [--steps 5]"""
import argparse, sys, time
import torch
sys.path.insert(0, '.')
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=4); ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--prompts", type=int, default=4); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
cfg = pkg.internvl2_8b()
B, T, P = a.clips, a.frames, a.prompts
N = synth.canonical_len(cfg, T)
dev = torch.device("cuda", 0)
model = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * (N + 8))
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
base = synth.canonical_tokens(cfg, B, T, seed=0)
model.img_context_token_id = base["img_context_token_id"]
model.eval()
prompts = synth.perspective_prompts(base, P, seed=0)
pv = synth.synthetic_frames(B * T, cfg.image_size, seed=0, device=dev)
motion = synth.synthetic_motion(B, cfg.motion_dim, seed=0, device=dev)
flags = torch.ones(B * T, 1, dtype=torch.long)

def separate():
    return [model(mos=None, pixel_values=pv, input_ids=p["input_ids"], attention_mask=p["attention_mask"], image_flags=flags,
                  labels=p["labels"], motion_feature=motion) for p in prompts]
def shared_vit():
    vt = model.vit_tokens(pv)
    return [model(mos=None, pixel_values=pv, visual_tokens=vt, input_ids=p["input_ids"], attention_mask=p["attention_mask"],
                  image_flags=flags, labels=p["labels"], motion_feature=motion) for p in prompts]
def shared_prefix():
    return model.forward_shared_prefix([(p["input_ids"], p["attention_mask"], p["labels"]) for p in prompts], pixel_values=pv,
                                       image_flags=flags, motion_feature=motion)
res = {}
for name, fn in (("one full pass per perspective (the reference's loop)", separate), ("ViT tokens shared", shared_vit),
                 ("video prefix shared (forward_shared_prefix)", shared_prefix)):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(a.steps): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / a.steps
    res[name] = out
    print(f"{name:58s}: {dt * 1e3:8.1f} ms per {B} clips x {P} perspectives = {B * P / dt:7.1f} scores/s", flush=True)
names = list(res)
# yardstick for the comparison below: the SAME separate pass, clip by clip instead of as a batch of B (other GEMM row bands,
# hence other fp32 summation orders - the bf16 noise floor of a 32-layer random-weight model, cf. tests/test_gpu_e2e.py)
single = []
for p in prompts:
    rows = [model(mos=None, pixel_values=pv[b * T:(b + 1) * T], input_ids=p["input_ids"][b:b + 1], attention_mask=p["attention_mask"][b:b + 1],
                  image_flags=flags[:T], labels=p["labels"][b:b + 1], motion_feature=motion[b:b + 1]) for b in range(B)]
    single.append({"score1": torch.cat([r["score1"] for r in rows]), "logit": torch.cat([r["logit"] for r in rows])})
for i in range(P):
    want = (prompts[i]["labels"][:, 1:] != -100).reshape(-1)
    ref, shp, one = res[names[0]][i], res[names[2]][i], single[i]
    d = (ref["score1"].float() - shp["score1"].float()).abs().max().item()
    dy = (ref["score1"].float() - one["score1"].float()).abs().max().item()
    n = int((ref["logit"].cpu()[want] != shp["logit"].cpu()[want]).sum()); ny = int((ref["logit"].cpu()[want] != one["logit"].cpu()[want]).sum())
    print(f"perspective {i}: separate vs shared prefix: max |d score1| {d:.4g}, level tokens differing {n}/{int(want.sum())}   "
          f"| yardstick, batch vs clip-by-clip separate passes: {dy:.4g}, {ny}/{int(want.sum())}")
