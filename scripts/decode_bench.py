"""generate() throughput on the 8B model: prefill + greedy decode steps (HBM-bound weight streaming).
python scripts/decode_bench.py [batch] [new tokens] [num_beams]   (num_beams > 1: HF beam search, all beams in one decode step per token)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
new = int(sys.argv[2]) if len(sys.argv) > 2 else 33
beams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
gen = dict(num_beams=beams) if beams > 1 else {}
cfg = pkg.internvl2_8b()
dev = torch.device('cuda', 0)
model = InternVLChatModel(cfg, device=dev, max_clips=B)
model.load_state_dict(synth.make_state_dict(cfg, seed=0, device=dev, rich=True))
if os.environ.get("PREC") == "fp8":            # fp8 mode of the InternLM2 linears: the decode GEMVs stream the e4m3 copies
    model.set_precision("fp8")
toks = synth.canonical_tokens(cfg, B, 8, seed=0)
model.img_context_token_id = toks["img_context_token_id"]
n_prompt = int((toks["labels"][0] == -100).sum())
ids = toks["input_ids"][:, :n_prompt].clone()
for b in range(B):
    ids[b, (ids[b] == model.img_context_token_id).nonzero()[-1]] = 7
pv = synth.synthetic_frames(B * 8, 448, seed=0, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=1, **gen)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = model.generate(pixel_values=pv, input_ids=ids, attention_mask=torch.ones_like(ids), max_new_tokens=new, **gen)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    per_tok = ((t2 - t1) - (t1 - t0)) / (new - 1)
    wbytes = sum(p.numel() for n, p in model.named_parameters() if n.startswith("language_model.model.layers") or n == "language_model.output.weight") * 2
    print(f"B={B} beams={beams} prefill+1 {1e3*(t1-t0):.1f} ms; decode {1e3*per_tok:.2f} ms/token ({B/per_tok:.1f} tok/s), weights streamed {wbytes/1e9:.1f} GB -> {wbytes/per_tok/1e12:.2f} TB/s", flush=True)
print(out[0].tolist()[:10])
