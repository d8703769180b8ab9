"""Soak run of the batched eval loop on the headline model (InternVL2-8B sizes, SlowFast branch inside): N clips - a small pool of uint8 720p clips and of ragged
prompts, cycled - through eval_utils.batched(k = 4) with graph replay on, checking what a long-running eval job needs: every repetition of a (clip, prompt) pair
gives the SAME bits as its first occurrence, device memory in use does not grow (torch allocator + the native context's own allocations via hipMemGetInfo), host RSS
does not grow, the graph cache stays bounded.

    python tests/manual/soak_loop.py [n_clips = 1200] [--uniform] [--k=K] [--tiny]       # MI355X, ~1 min (--k: group size of the loop, default 4)"""
import os
import resource
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import eval_utils, synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402
from aigv_assessor_amd.slowfast import SlowFastR50  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_clips = int(args[0]) if args else 1200
tiny = "--tiny" in sys.argv                                            # a small model (debugging the loop itself)
uniform = "--uniform" in sys.argv                                      # every clip behind the same prompt (the reference's eval set per perspective): every group replays
K = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--k=")), 4)
cfg = pkg.tiny(image_size=224, vit_layers=2, llm_layers=2) if tiny else pkg.internvl2_8b()
T = 8
dev = torch.device("cuda", 0)
N = synth.canonical_len(cfg, T)
model = InternVLChatModel(cfg, device=dev, max_clips=K, max_frames=K * T, max_tokens=K * (N + 8))
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
model.eval()
model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=0))
model.enable_graph_replay(True)
g = torch.Generator().manual_seed(11)
pool = [torch.randint(0, 256, (T, 720, 1280, 3) if not tiny else (T, 240, 320, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(5)]
prompts = []
for p in range(3):                                                   # three prompt lengths (N, N + 2, N + 5): the loop's groups are ragged
    toks = synth.canonical_tokens(cfg, 1, T, seed=p)
    ids, lab = toks["input_ids"], toks["labels"]
    extra = (0, 2, 5)[p]
    if extra:
        a0 = int((lab[0] != -100).nonzero()[0])
        ids = torch.cat([ids[:, :a0], torch.randint(3, min(90000, cfg.llm_config.vocab_size - 600), (1, extra), generator=g), ids[:, a0:]], 1)
        lab = torch.cat([lab[:, :a0], torch.full((1, extra), -100), lab[:, a0:]], 1)
    prompts.append((ids, lab))
model.img_context_token_id = toks["img_context_token_id"]


def items():
    for i in range(n_clips):
        ids, lab = prompts[0] if uniform else prompts[(i // K) % 3 if i % 7 else i % 3]   # mostly group-uniform prompts (graphs replay), now and then a ragged group (eager)
        yield {"input_ids": ids, "labels": lab, "attention_mask": torch.ones_like(ids, dtype=torch.bool), "image_flags": torch.ones(1, T, 1, dtype=torch.long),
               "frames": pool[i % 5], "key": (i % 5, ids.shape[1])}


def mem():
    free, total = torch.cuda.mem_get_info(dev)
    return (total - free) / 2 ** 20, torch.cuda.memory_allocated(dev) / 2 ** 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024


first, mismatches, marks = {}, 0, []
t0 = time.time()
for j, (it, out) in enumerate(eval_utils.batched(items(), model, k=K, frames=lambda it: it["frames"])):
    val = (out["score1"].item(), tuple(eval_utils.answer_ids(it["labels"][0], out["logit"]).tolist()))
    if first.setdefault(it["key"], val) != val:
        mismatches += 1
    if j in (199, n_clips // 2, n_clips - 1):
        torch.cuda.synchronize()
        marks.append((j + 1,) + mem())
dt = time.time() - t0
print(f"{n_clips} clips in {dt:.1f} s = {n_clips / dt:.1f} clips/s through eval_utils.batched(k = {K}) (uint8 720p frames from pinned host memory, graph replay on, {'one prompt for every clip' if uniform else 'ragged groups mixed in'})")
print(f"{len(first)} distinct (clip, prompt length) pairs; repetitions that differ from their first occurrence: {mismatches}")
for n, used, alloc, rss in marks:
    print(f"after {n:5d} clips: device memory in use {used:9.0f} MiB (torch allocator {alloc:9.0f} MiB), host max RSS {rss:7.0f} MiB")
print(f"graph cache entries: {len(model._graphs)} (bound {model.GRAPH_CACHE_SIZE})")
grow = marks[-1][1] - marks[0][1]
assert mismatches == 0 and grow < 64 and len(model._graphs) <= model.GRAPH_CACHE_SIZE, (mismatches, grow)
print("SOAK_OK")
