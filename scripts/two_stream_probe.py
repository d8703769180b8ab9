"""Probe: do two independent scoring passes on two HIP streams (two contexts, own workspaces) finish sooner than back to back?
If the launches of one pass fill the idle tails (partial tile rounds, causal attention drain, small memory-bound kernels) of the
other, aggregate clips/s rises; if the MFMA-bound kernels merely share the chip, it does not.   python scripts/two_stream_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
cfg = pkg.internvl2_8b()
dev = torch.device("cuda", 0)
B, T = 4, 8
N = synth.canonical_len(cfg, T)
sd = synth.make_state_dict(cfg, seed=0, device=dev, rich=True)
models = []
for i in range(2):
    m = InternVLChatModel(cfg, device=dev, max_clips=B, max_frames=B * T, max_tokens=B * N)
    m.load_state_dict(sd)
    m.eval()
    models.append(m)
del sd
toks = synth.canonical_tokens(cfg, B, T, seed=0)
for m in models:
    m.img_context_token_id = toks["img_context_token_id"]
pv = synth.synthetic_frames(B * T, 448, seed=0, device=dev)
motion = synth.synthetic_motion(B, cfg.motion_dim, seed=0, device=dev)
flags = torch.ones(B * T, 1, dtype=torch.long)
kw = dict(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"], motion_feature=motion)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def run(n_steps, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps):
        k = i % 2 if two else 0
        with torch.cuda.stream(streams[k]):
            out = models[k](**kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_steps
for _ in range(2):
    run(4, True)
for rep in range(3):
    a = run(12, False)
    b = run(12, True)
    print(f"one stream {1e3 * a:.1f} ms/step = {B / a:.2f} clips/s;  two streams alternating {1e3 * b:.1f} ms/step = {B / b:.2f} clips/s  ({100 * (a / b - 1):+.1f} %)", flush=True)
