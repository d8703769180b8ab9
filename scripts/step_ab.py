"""In-process A/B of the benched step (InternVL2-8B, 4 clips x 8 frames x 448 px, SlowFast branch inside) over context knobs:

    python scripts/step_ab.py knob=value[,knob=value...] [...]      e.g.  tiny_side=0 tiny_side=1

Every argument is one arm (InternVLChatModel.tune knobs, or attr.NAME=value for a Python-side switch of the model); arms are interleaved over four rounds in ONE process, each visit = 2 untimed + 8 timed eager
steps between HIP events.  Per arm: median / min ms per step, and whether scores and level tokens equal the first arm's bit for bit."""
import sys
sys.path.insert(0, ".")
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50

arms = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[1:]] or [{}]
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
pv = synth.synthetic_frames(B * T, 448, seed=0).to(dev)
flags = torch.ones(B * T, 1, dtype=torch.long)


def step():
    return model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"])


res = {i: [] for i in range(len(arms))}
outs = {}
for rnd in range(4):
    for ai, arm in enumerate(arms):
        for k, v in arm.items():
            if k.startswith("attr."):           # a Python-side switch of the model (e.g. attr.gate_motion_branch=1)
                setattr(model, k[5:], v)
                model._drop_graphs()
            else:
                model.tune(k, v)
        for _ in range(2):
            o = step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            o = step()
        e1.record()
        torch.cuda.synchronize()
        res[ai].append(e0.elapsed_time(e1) / 8)
        outs[ai] = (o["score1"].clone(), o["logit"].clone())
        for k in arm:
            if k.startswith("attr."):
                setattr(model, k[5:], 0)
                model._drop_graphs()
            else:
                model.tune(k, -1)
for ai, arm in enumerate(arms):
    v = sorted(res[ai])
    same = torch.equal(outs[ai][0], outs[0][0]) and torch.equal(outs[ai][1], outs[0][1])
    print(f"{str(arm):40s} ms/step median {(v[1] + v[2]) / 2:7.2f} (min {v[0]:.2f}, max {v[-1]:.2f})   bits equal to arm 0: {same}", flush=True)
