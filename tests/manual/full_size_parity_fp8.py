"""Full-size check of the fp8 mode (InternVL2-8B widths and depths, one 8-frame 448x448 clip): the HIP path in fp8 mode against
oracle/fp8.py (the same definition evaluated on the CPU, all 32 layers), with the bf16 oracle as the yardstick for what the mode costs.
Run by hand on the GPU box:   python tests/manual/full_size_parity_fp8.py [seed]      (result of round 1: profiles/parity_fp8_mode_r1.txt)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
from oracle import oracle as O
from oracle import fp8 as O8

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = pkg.internvl2_8b()
B, T = 1, 8
sd = synth.make_state_dict(cfg, seed=seed, rich=True)
toks = synth.canonical_tokens(cfg, B, T, seed=seed)
pv = synth.synthetic_frames(B * T, cfg.image_size, seed=seed)
motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
flags = torch.ones(B * T, 1, dtype=torch.long)
args = (sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion, toks["img_context_token_id"])
t0 = time.time()
with torch.no_grad():
    ref16 = O.forward_eval(*args, mos=None, stage=2, return_intermediates=True)
    t1 = time.time()
    with O8.fp8_llm(cfg.llm_config.num_hidden_layers):
        ref8 = O.forward_eval(*args, mos=None, stage=2, return_intermediates=True)
print(f"seed {seed}: bf16 oracle {t1 - t0:.0f} s, fp8 oracle {time.time() - t1:.0f} s ({torch.get_num_threads()} threads)", flush=True)
model = InternVLChatModel(cfg, device=torch.device("cuda", 0), max_clips=B, max_frames=B * T, max_tokens=B * toks["input_ids"].shape[1])
model.load_state_dict(sd)
model.img_context_token_id = toks["img_context_token_id"]
model.eval()
kw = dict(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"],
          motion_feature=motion)
model.set_precision("fp8")
out8 = model(**kw)
all8 = model(full_logits=True, **kw)["logit"].cpu()
model.set_precision("bf16")
out16 = model(**kw)
torch.cuda.synchronize()
want = ref8["label"] != -100
h16, h8 = ref16["hidden"].float(), ref8["hidden"].float()
s = lambda o: o["score1"].float().cpu().item()
print(f"score1: hip fp8 {s(out8):.5f} | fp8 oracle {s(ref8):.5f} | hip bf16 {s(out16):.5f} | bf16 oracle {s(ref16):.5f}")
print(f"|hip fp8 - fp8 oracle| = {abs(s(out8) - s(ref8)):.5f}   vs the price of the mode |fp8 oracle - bf16 oracle| = {abs(s(ref8) - s(ref16)):.5f}")
print(f"answer-row level tokens: hip fp8 == fp8 oracle {int((out8['logit'].cpu()[want] == ref8['logit'][want]).sum())}/{int(want.sum())}; "
      f"fp8 oracle == bf16 oracle {int((ref8['logit'][want] == ref16['logit'][want]).sum())}/{int(want.sum())}")
print(f"all-row argmax agreement: hip fp8 vs fp8 oracle {float((all8 == ref8['logit']).float().mean()):.3f}; fp8 oracle vs bf16 oracle "
      f"{float((ref8['logit'] == ref16['logit']).float().mean()):.3f}")
print(f"final hidden state, fp8 oracle vs bf16 oracle: relative L2 distance {float((h8 - h16).norm() / h16.norm()):.4f}")
