"""In-process A/B of the reference's own eval-loop shape (batch 1 + ingest + per-clip host sync; bench.py reference_loop_metric) on the 8B scorer:

    python scripts/loop_shape_ab.py knob=value[,knob=value...] [...]      e.g.  co_kmax=0 co_kmax=1024 co_kmax=4096 co_kmax=16384

Every argument is one arm (context knobs of InternVLChatModel.tune); arms are interleaved over three rounds in ONE process.  Per arm: median
wall ms per clip, and the GPU time of the forward alone (HIP events around model(...))."""
import sys, time
sys.path.insert(0, ".")
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import eval_utils, synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50

arms = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[1:]] or [{}]
cfg = pkg.internvl2_8b()
T = 8
dev = torch.device("cuda", 0)
N = synth.canonical_len(cfg, T)
model = InternVLChatModel(cfg, device=dev, max_clips=4, max_frames=4 * T, max_tokens=4 * N)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
toks = synth.canonical_tokens(cfg, 4, T, seed=0)
model.img_context_token_id = toks["img_context_token_id"]
model.eval()
model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=0))
g8 = torch.Generator().manual_seed(6)
clips = [torch.randint(0, 256, (T, 720, 1280, 3), dtype=torch.uint8, generator=g8).pin_memory() for _ in range(2)]
ids, labels, am = toks["input_ids"][:1], toks["labels"][:1], toks["attention_mask"][:1]
flags = torch.ones(T, 1, dtype=torch.long)


def one(i):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pv = model.ingest_frames(clips[i % 2].to(dev, non_blocking=True))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = model(mos=None, pixel_values=pv, input_ids=ids, attention_mask=am, image_flags=flags, labels=labels, motion_feature=None)
    e1.record()
    score = out["score1"].item()
    eval_utils.answer_ids(labels[0], out["logit"].cpu())
    wall = (time.perf_counter() - t0) * 1e3
    return wall, e0.elapsed_time(e1), score


res = {i: [] for i in range(len(arms))}
for rnd in range(3):
    for ai, arm in enumerate(arms):
        for k, v in arm.items():
            model.tune(k, v)
        for i in range(8):
            w, gq, sc = one(i)
            if i >= 2:
                res[ai].append((w, gq, sc))
        for k in arm:
            model.tune(k, -1)
for ai, arm in enumerate(arms):
    ws = sorted(r[0] for r in res[ai]); gs = sorted(r[1] for r in res[ai])
    print(f"{str(arm):40s} wall ms/clip median {ws[len(ws) // 2]:7.2f} (min {ws[0]:.2f})   forward GPU ms median {gs[len(gs) // 2]:7.2f}   score {res[ai][0][2]:.6f}", flush=True)
