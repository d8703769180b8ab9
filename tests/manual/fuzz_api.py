"""Randomised differential run of the model's API under graph replay (tiny model, one MI355X): a device-under-test with `enable_graph_replay()` on and whatever state the
operations before left behind (captured graphs, grown contexts, cached SlowFast handles, a prefetch in flight) against a FRESH eager model per operation.  Operations:
forward at 1-4 clips x 8 / 12 / 16 frames (SlowFast inside or a given motion feature; stage 2), the four-perspective shared-prefix pass, generate() at 1-3 sequences,
the look-ahead loop, eval_utils.batched, score_clips_dp on one rank, frame ingest at several decoded sizes, extract_feature / vit_tokens / motion_feature, beam search, weight reloads on the live model, up to three long-lived models sharing one
SlowFast branch, and mode toggles (precision bf16 / fp8, attention numerics, row
trimming, GEMM mode) applied to both sides.  Every result must equal the fresh model's bit for bit.  (Parity with the reference is other tests' business: this one hunts state.)

    python tests/manual/fuzz_api.py [n_ops = 150] [seed = 0]"""
import faulthandler
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import dist_utils, eval_utils, synth  # noqa: E402
from aigv_assessor_amd.modeling import InternVLChatModel  # noqa: E402
from aigv_assessor_amd.slowfast import SlowFastR50  # noqa: E402

faulthandler.enable()
if os.environ.get("FUZZ_NO_RETIRE"):      # diagnosis: dropped graphs stay parked for the life of the process
    from aigv_assessor_amd import modeling as _modeling
    _modeling._retire_parked_graphs = lambda *a, **k: 0
VERBOSE = bool(os.environ.get("FUZZ_VERBOSE"))
BIG = "--8b" in sys.argv          # InternVL2-8B sizes (weights generated on the device; the eager comparison model is then built once and reused)
sys.argv = [a for a in sys.argv if a != "--8b"]
n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = pkg.internvl2_8b() if BIG else pkg.tiny(image_size=224, vit_layers=2, llm_layers=2)
S = cfg.image_size
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(cfg, seed=5, rich=True, **({"device": dev} if BIG else {}))
print(f"model: InternViT {cfg.vision_config.hidden_size} x {cfg.vision_config.num_hidden_layers}, LLM {cfg.llm_config.hidden_size} x {cfg.llm_config.num_hidden_layers}, {S} px", flush=True)
_REF = []
sf_sd = synth.slowfast_state_dict(seed=3)
modes = dict(precision="bf16", numerics="fp32", trim=True, gemm=-1)


def make(graph):
    if BIG and not graph and _REF:            # the eager comparison model: one for the whole run at these sizes (still eager, still without captured graphs)
        apply_modes(_REF[0])
        return _REF[0]
    m = InternVLChatModel(cfg, device=dev, max_clips=1)
    m.load_state_dict(sd)
    m.eval()
    m.slowfast_model = SlowFastR50(sf_sd)
    apply_modes(m)
    m.enable_graph_replay(graph)
    if BIG and not graph:
        _REF.append(m)
    return m


def apply_modes(m):
    m.set_precision(modes["precision"])
    m.set_attention_numerics(modes["numerics"])
    m.set_row_trimming(modes["trim"])
    m.set_gemm_mode(modes["gemm"])


dut = make(True)
duts = [dut]
rng = random.Random(seed0)
g = torch.Generator().manual_seed(seed0)
bad, counts = 0, {}


def same(a, b):
    if torch.is_tensor(a):
        return torch.is_tensor(b) and a.shape == b.shape and torch.equal(a.detach().cpu(), b.detach().cpu())
    if isinstance(a, dict):
        return all(same(a[k], b[k]) for k in a if torch.is_tensor(a[k]))
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return a == b


def inputs(B, T, seed, branch, extra=0):
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    ids, lab = toks["input_ids"], toks["labels"]
    if extra:
        a0 = int((lab[0] != -100).nonzero()[0])
        fill = torch.randint(3, cfg.llm_config.vocab_size - 16, (B, extra), generator=torch.Generator().manual_seed(seed))
        ids = torch.cat([ids[:, :a0], fill, ids[:, a0:]], 1)
        lab = torch.cat([lab[:, :a0], torch.full((B, extra), -100), lab[:, a0:]], 1)
    kw = dict(mos=None, pixel_values=synth.synthetic_frames(B * T, S, seed=seed).to(dev), input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool),
              image_flags=torch.ones(B * T, 1, dtype=torch.long), labels=lab)
    if not branch:
        kw["motion_feature"] = synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(dev)
    return kw, toks["img_context_token_id"]


def op_forward(m, B, T, seed, branch, extra):
    kw, ctx = inputs(B, T, seed, branch, extra)
    m.img_context_token_id = ctx
    return m(**kw)


def op_prefix(m, B, T, seed):
    base = synth.canonical_tokens(cfg, B, T, seed=seed)
    m.img_context_token_id = base["img_context_token_id"]
    prompts = [(p["input_ids"], p["attention_mask"], p["labels"]) for p in synth.perspective_prompts(base, 4, seed=seed)]
    outs = m.forward_shared_prefix(prompts, pixel_values=synth.synthetic_frames(B * T, S, seed=seed).to(dev), image_flags=torch.ones(B * T, 1, dtype=torch.long))
    return [{k: v for k, v in o.items() if torch.is_tensor(v)} for o in outs]


def op_generate(m, B, T, n_new, seed, beams=1, extra=None):
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    ctx = toks["img_context_token_id"]
    m.img_context_token_id = ctx
    n_prompt = int((toks["labels"][0] == -100).sum())
    ids = toks["input_ids"][:, :n_prompt].clone()
    for b in range(B):
        ids[b, (ids[b] == ctx).nonzero()[-1]] = 7
    return m.generate(pixel_values=synth.synthetic_frames(B * T, S, seed=seed).to(dev), input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=n_new,
                      **{"do_sample": False, **({} if beams == 1 else {"num_beams": beams}), **_gen_extra(extra, seed)})


def _gen_extra(extra, seed):
    """Sampling (a fresh seeded device generator per call: the same draws on both sides when the logits are the same bits), an EOS id, HF's logits processors."""
    if not extra:
        return {}
    kw = dict(extra)
    if kw.pop("sample", False):
        kw.update(do_sample=True, generator=torch.Generator(device=dev).manual_seed(1000 + seed))
    return kw


def loop_items(n, T, hw, seed):
    gg = torch.Generator().manual_seed(seed)
    items, ctx = [], None
    for i in range(n):
        toks = synth.canonical_tokens(cfg, 1, T, seed=seed + i)
        ctx = toks["img_context_token_id"]
        items.append({"input_ids": toks["input_ids"], "labels": toks["labels"], "attention_mask": toks["attention_mask"], "image_flags": torch.ones(1, T, 1, dtype=torch.long),
                      "frames": torch.randint(0, 256, (T,) + hw + (3,), dtype=torch.uint8, generator=gg).pin_memory()})
    return items, ctx


def op_batched(m, n, T, hw, k, ahead, seed):
    items, ctx = loop_items(n, T, hw, seed)
    m.img_context_token_id = ctx
    return [o for _, o in eval_utils.batched(items, m, k=k, ahead=ahead, frames=lambda it: it["frames"])]


def op_lookahead(m, n, T, hw, seed):
    items, ctx = loop_items(n, T, hw, seed)
    m.img_context_token_id = ctx
    outs = []
    for it, ahead in eval_utils.lookahead(items, m, frames=lambda it: it["frames"]):
        o = m(mos=None, pixel_values=ahead, input_ids=it["input_ids"], attention_mask=it["attention_mask"], image_flags=it["image_flags"][0], labels=it["labels"])
        outs.append({k: v.clone() for k, v in o.items() if torch.is_tensor(v)})
    return outs


def op_dp(m, B, T, seed, branch):
    kw, ctx = inputs(B, T, seed, branch)
    m.img_context_token_id = ctx
    return dist_utils.score_clips_dp(m, kw["pixel_values"], kw["input_ids"], kw["attention_mask"], kw["image_flags"], kw["labels"], kw.get("motion_feature"))


def op_feature(m, F, seed):
    pv = synth.synthetic_frames(F, S, seed=seed).to(dev)
    return [m.extract_feature(pv), m.vit_tokens(pv), m.motion_feature(synth.synthetic_frames(8, S, seed=seed).to(dev), 1)]


def op_ingest(m, T, hw, seed):
    gg = torch.Generator().manual_seed(seed)
    return m.ingest_frames(torch.randint(0, 256, (T,) + hw + (3,), dtype=torch.uint8, generator=gg).to(dev))


for it in range(n_ops):
    r = rng.random()
    T = rng.choice([8, 8, 12, 16])
    if r < 0.30:
        name, fn = "forward", (lambda m, a=(rng.randint(1, 4), T, rng.randint(0, 3), rng.random() < 0.6, rng.choice([0, 0, 2])): op_forward(m, *a))
    elif r < 0.40:
        name, fn = "prefix", (lambda m, a=(rng.randint(1, 2), T, rng.randint(0, 2)): op_prefix(m, *a))
    elif r < 0.52:
        name, fn = "generate", (lambda m, a=(rng.randint(1, 3), T, rng.randint(2, 6), rng.randint(0, 2), rng.choice([1, 1, 3])), e=rng.choice([None, None, dict(sample=True, temperature=0.8, top_k=20, top_p=0.9), dict(sample=True),
            dict(repetition_penalty=1.3, no_repeat_ngram_size=2), dict(eos_token_id=[5, 9, 11, 17, 23, 42, 77, 101, 300, 301, 302, 303])]): op_generate(m, *(a if not (e and e.get("sample")) else a[:4] + (1,)), extra=e))
    elif r < 0.66:
        name, fn = "batched", (lambda m, a=(rng.randint(1, 7), T, rng.choice([(240, 320), (300, 400), (224, 224)]), rng.randint(1, 4), rng.random() < 0.7, rng.randint(0, 2)): op_batched(m, *a))
    elif r < 0.74:
        name, fn = "lookahead", (lambda m, a=(rng.randint(1, 4), T, rng.choice([(240, 320), (300, 400)]), rng.randint(0, 2)): op_lookahead(m, *a))
    elif r < 0.84:
        name, fn = "dp", (lambda m, a=(rng.randint(1, 4), T, rng.randint(0, 3), rng.random() < 0.6): op_dp(m, *a))
    elif r < 0.87:
        name, fn = "feature", (lambda m, a=(rng.choice([1, 3, 8, 20]), rng.randint(0, 3)): op_feature(m, *a))
    elif r < 0.90:
        name, fn = "ingest", (lambda m, a=(rng.choice([4, 8]), rng.choice([(240, 320), (360, 640), (224, 224)]), rng.randint(0, 3)): op_ingest(m, *a))
    elif r < 0.93:
        # the weights loaded again on the live model (same values: results must not move; the captured graphs go), or the model replaced by a second long-lived one
        # that shares the SlowFast branch with the first (its geometry churn retires handles under the other's graphs)
        if rng.random() < 0.5:
            dut.load_state_dict(sd)
        else:
            duts.append(make(True))
            duts[-1].slowfast_model = duts[0].slowfast_model
            if len(duts) > (2 if BIG else 3):      # (an 8B model holds ~52 GB: parameters, the native copy, workspaces - three of them, the comparison model and the weights' source fill the card)
                duts.pop(1)
                import gc
                gc.collect()
        counts["reload/second"] = counts.get("reload/second", 0) + 1
        continue
    else:
        which = rng.choice(["precision", "numerics", "trim", "gemm"])
        modes[which] = {"precision": rng.choice(["bf16", "fp8"]), "numerics": rng.choice(["fp32", "reference"]), "trim": rng.random() < 0.7, "gemm": rng.choice([-1, -1, 2])}[which]
        for d in duts:
            apply_modes(d)
        counts["toggle"] = counts.get("toggle", 0) + 1
        continue
    counts[name] = counts.get(name, 0) + 1
    if VERBOSE:
        print(f"op {it} {name} {fn.__defaults__} on dut {it % len(duts)} of {len(duts)}", flush=True)
    ref = make(False)
    want = fn(ref)
    reps = 3 if name in ("forward", "dp") else 1          # (repeat: eager -> captured -> replayed)
    for rep in range(reps):
        got = fn(duts[(it + rep) % len(duts)])
        if not same(want, got):
            bad += 1
            print(f"MISMATCH op {it} {name} rep {rep} modes {modes}", flush=True)
    del ref, want, got
    if it % 25 == 24:
        print(f"op {it + 1}/{n_ops}: {counts}; mismatches {bad}; captured graphs {sum(isinstance(v, tuple) for v in dut._graphs.values())}", flush=True)
assert bad == 0, bad
from aigv_assessor_amd import modeling as _m  # noqa: E402
free, total = torch.cuda.mem_get_info(dev)
print(f"dropped graphs parked: {len(_m._PARKED_GRAPHS)} (limit {InternVLChatModel.PARKED_GRAPHS_LIMIT}); device memory in use {(total - free) / 2 ** 30:.1f} GiB")
print(f"FUZZ_API_OK {n_ops} ops {counts}")
