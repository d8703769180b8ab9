import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aigv_assessor_amd as pkg
from aigv_assessor_amd import eval_utils, synth
from aigv_assessor_amd.modeling import InternVLChatModel
from aigv_assessor_amd.slowfast import SlowFastR50
big = "--8b" in sys.argv
cfg = pkg.internvl2_8b() if big else pkg.tiny(image_size=224, vit_layers=2, llm_layers=2)
T = 8
dev = torch.device("cuda", 0)
model = InternVLChatModel(cfg, device=dev, max_clips=4)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
model.eval()
model.slowfast_model = SlowFastR50(synth.slowfast_state_dict(seed=0))
model.enable_graph_replay(True)
toks = synth.canonical_tokens(cfg, 1, T, seed=0)
model.img_context_token_id = toks["img_context_token_id"]
g = torch.Generator().manual_seed(1)
hw = (720, 1280) if big else (240, 320)
pool = [torch.randint(0, 256, (T,) + hw + (3,), dtype=torch.uint8, generator=g).pin_memory() for _ in range(3)]
def items(n):
    for i in range(n):
        yield {"input_ids": toks["input_ids"], "labels": toks["labels"], "attention_mask": toks["attention_mask"], "image_flags": torch.ones(1, T, 1, dtype=torch.long), "frames": pool[i % 3], "i": i}
first = {}
bad = 0
for rnd in range(6):
    for it, out in eval_utils.batched(items(24), model, k=4, frames=lambda it: it["frames"]):
        v = out["score1"].item()
        if first.setdefault(it["i"] % 3, v) != v: bad += 1
    n = sum(isinstance(v, tuple) for v in model._graphs.values())
    # drop every captured graph in three different ways, then run again (recapture under the look-ahead)
    [lambda: model.set_gemm_mode(-1), lambda: model._drop_graphs(), lambda: model.set_attention_numerics("fp32")][rnd % 3]()
    if rnd in (1, 4):     # a generate() call between the rounds: the KV cache is sized (a context resize: graphs dropped), decode steps run eager, then the loop recaptures
        n_prompt = int((toks["labels"][0] == -100).sum())
        ids = toks["input_ids"][:, :n_prompt].clone()
        ids[0, (ids[0] == toks["img_context_token_id"]).nonzero()[-1]] = 7
        pvg = model.ingest_frames(pool[0].cuda())
        gen = model.generate(pixel_values=pvg, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=4 + rnd, do_sample=False).cpu()
        if "gen" in first and first["gen"] != gen[0, :4].tolist(): bad += 1
        first.setdefault("gen", gen[0, :4].tolist())
    print(f"round {rnd}: captured graphs before the drop {n}, mismatches so far {bad}", flush=True)
# a bigger group mid-run: the context grows (resize), graphs are dropped, recaptured
model2_items = list(items(16))
for it, out in eval_utils.batched(model2_items, model, k=8, frames=lambda it: it["frames"]):
    v = out["score1"].item()
    if first.setdefault(it["i"] % 3, v) != v: bad += 1
for it, out in eval_utils.batched(items(24), model, k=4, frames=lambda it: it["frames"]):
    v = out["score1"].item()
    if first.setdefault(it["i"] % 3, v) != v: bad += 1
print("DROP_RECAPTURE_OK" if bad == 0 else f"MISMATCHES {bad}", sum(isinstance(v, tuple) for v in model._graphs.values()))
