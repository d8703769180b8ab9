"""Does the scorer's forward capture into a HIP graph (torch.cuda.graph around InternVLChatModel.forward: ~1000 launches through the C ABI on
torch's current stream + the SlowFast side stream)?  Compares replay with eager bit for bit and times both.   python scripts/graph_probe.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50
cfg = pkg.internvl2_8b()
B, T = 4, 8
dev = torch.device("cuda", 0)
N = synth.canonical_len(cfg, T)
model = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * N)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
toks = synth.canonical_tokens(cfg, B, T, seed=0)
model.img_context_token_id = toks["img_context_token_id"]
model.eval()
model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=0))
pv = synth.synthetic_frames(B * T, cfg.image_size, seed=0).to(dev)
flags = torch.ones(B * T, 1, dtype=torch.long)
step = lambda: model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"])
for _ in range(3):
    eager = step()
torch.cuda.synchronize()
try:
    replay, cap = model.capture_forward(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"])
except Exception as e:
    print("capture failed:", repr(e)[:600])
    raise SystemExit(1)
replay(); torch.cuda.synchronize()
print("replay == eager: score", torch.equal(cap["score1"], eager["score1"]), "logit", torch.equal(cap["logit"], eager["logit"]), cap["score1"].tolist())
pv2 = synth.synthetic_frames(B * T, cfg.image_size, seed=1).to(dev)
e2 = model(mos=None, pixel_values=pv2, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"])
want2 = e2["score1"].clone(); torch.cuda.synchronize()
pv.copy_(pv2); replay(); torch.cuda.synchronize()
print("other frames through the same graph == eager:", torch.equal(cap["score1"], want2), cap["score1"].tolist())
for name, fn in (("eager", step), ("graph replay", replay)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:14s} {(t2 - t0) / 20 * 1e3:8.2f} ms/step   host enqueue {(t1 - t0) / 20 * 1e3:7.3f} ms/step", flush=True)
