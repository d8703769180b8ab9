"""Mixed-workload soak on the headline model (InternVL2-8B sizes, SlowFast inside, graph replay on): a seeded random sequence of everything a user calls - forward at 1 / 2 / 3 / 5 clips,
the four-perspective shared-prefix pass, generate() at one and two sequences with different lengths, frame ingest at two resolutions, a mode toggle now and then - where every operation
with the same arguments must return the same bits as the first time, device memory must not grow, and nothing may raise.

    python tests/manual/soak_mixed.py [iterations = 300] [--tiny]        # MI355X"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402
from aigv_assessor_amd.slowfast import SlowFastR50  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
iters = int(args[0]) if args else 300
tiny = "--tiny" in sys.argv
cfg = pkg.tiny(image_size=224, vit_layers=2, llm_layers=2) if tiny else pkg.internvl2_8b()
S = cfg.image_size
T = 8
dev = torch.device("cuda", 0)
model = InternVLChatModel(cfg, device=dev, max_clips=2)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
model.eval()
model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=0))
model.enable_graph_replay(True)
g = torch.Generator().manual_seed(3)
u8 = {hw: [torch.randint(0, 256, (T,) + hw + (3,), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)] for hw in ((240, 320), (360, 640))}
first, counts, mism = {}, {}, 0
rng = random.Random(7)


def check(key, *tensors):
    global mism
    val = tuple(t.detach().cpu().reshape(-1).tolist() for t in tensors)
    counts[key[0]] = counts.get(key[0], 0) + 1
    if first.setdefault(key, val) != val:
        mism += 1
        print("MISMATCH", key, flush=True)


def frames(B, seed):
    return synth.synthetic_frames(B * T, S, seed=seed).to(dev)


def op_forward(B, seed):
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    model.img_context_token_id = toks["img_context_token_id"]
    o = model(mos=None, pixel_values=frames(B, seed), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=toks["labels"])
    check(("forward", B, seed), o["score1"], o["logit"])


def op_prefix(seed):
    B = 2
    base = synth.canonical_tokens(cfg, B, T, seed=seed)
    model.img_context_token_id = base["img_context_token_id"]
    prompts = [(p["input_ids"], p["attention_mask"], p["labels"]) for p in synth.perspective_prompts(base, 4, seed=seed)]
    outs = model.forward_shared_prefix(prompts, pixel_values=frames(B, seed), image_flags=torch.ones(B * T, 1, dtype=torch.long))
    check(("prefix", seed), *[o["score1"] for o in outs], *[o["logit"] for o in outs])


def op_generate(B, n_new, seed):
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    ctx = toks["img_context_token_id"]
    model.img_context_token_id = ctx
    n_prompt = int((toks["labels"][0] == -100).sum())
    ids = toks["input_ids"][:, :n_prompt].clone()
    for b in range(B):
        ids[b, (ids[b] == ctx).nonzero()[-1]] = 7
    out = model.generate(pixel_values=frames(B, seed), input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=n_new, do_sample=False)
    check(("generate", B, n_new, seed), out)


def op_ingest(hw, i):
    check(("ingest", hw, i), model.ingest_frames(u8[hw][i].to(dev, non_blocking=True)).float().sum())


t0 = time.time()
marks = []
for it in range(iters):
    r = rng.random()
    if r < 0.45:
        op_forward(rng.choice((1, 2, 3, 5)), rng.choice((0, 1)))
    elif r < 0.60:
        op_prefix(rng.choice((0, 1)))
    elif r < 0.80:
        op_generate(rng.choice((1, 2)), rng.choice((3, 7)), rng.choice((0, 1)))
    elif r < 0.95:
        op_ingest(rng.choice(((240, 320), (360, 640))), rng.choice((0, 1)))
    else:   # a mode toggle and back: drops every captured graph (behind a device synchronisation); the results must not move
        model.set_gemm_mode(2)
        model.set_gemm_mode(-1)
        counts["toggle"] = counts.get("toggle", 0) + 1
    if it in (iters // 8, iters // 4, iters // 2, 3 * iters // 4, iters - 1):
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info(dev)
        marks.append((it + 1, (total - free) / 2 ** 20, torch.cuda.memory_reserved(dev) / 2 ** 20))
print(f"{iters} operations in {time.time() - t0:.1f} s: {counts}; {len(first)} distinct (operation, arguments) keys; repetitions that differ from their first occurrence: {mism}")
print("device memory in use / of which torch's caching allocator holds (MiB):", ", ".join(f"after {n}: {m:.0f} / {r:.0f}" for n, m, r in marks),
      f"; captured graphs at the end: {sum(isinstance(v, tuple) for v in model._graphs.values())}")
outside = [m - r for _n, m, r in marks]          # what the native contexts and the runtime hold; the caching allocator's pool may still grow towards its plateau.  Every toggle
# drops the captured graphs, and dropped graph objects are PARKED, never destroyed (modeling._drop_graphs): ~2 MiB of runtime memory per parked graph - bounded per drop
per_toggle = (outside[-1] - outside[1]) / max(counts.get("toggle", 0), 1)
print(f"memory outside torch's allocator (MiB): {[round(x) for x in outside]}; growth per graph drop {per_toggle:.1f} MiB ({counts.get('toggle', 0)} drops)")
assert mism == 0 and outside[-1] - outside[1] < 64 + 6 * counts.get("toggle", 0), outside
print("MIXED_SOAK_OK")
