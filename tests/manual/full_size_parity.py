"""Full-size parity run (InternVL2-8B widths and depths, one 8-frame 448x448 clip, N = 2177): the CPU oracle (torch bf16 eager
restatement of the reference path, all 24 + 32 layers, ~1-2 min on the GPU box's host cores) against the HIP path on the same
seeded weights and inputs.  Too slow for the test suite; run by hand:   python tests/manual/full_size_parity.py [seed]
Result of round 1: profiles/parity_full_size_r1.txt"""
import sys, time
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
from oracle import oracle as O

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = pkg.internvl2_8b()
B, T = 1, 8
torch.set_num_threads(max(1, torch.get_num_threads()))
t0 = time.time()
sd = synth.make_state_dict(cfg, seed=seed, rich=True)                      # bf16 on the host, 16 GB
toks = synth.canonical_tokens(cfg, B, T, seed=seed)
pv = synth.synthetic_frames(B * T, cfg.image_size, seed=seed)
motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
flags = torch.ones(B * T, 1, dtype=torch.long)
print(f"weights + inputs generated in {time.time() - t0:.0f} s ({torch.get_num_threads()} threads)", flush=True)
t0 = time.time()
with torch.no_grad():
    ref = O.forward_eval(sd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion,
                         toks["img_context_token_id"], mos=None, stage=2, return_intermediates=True)
print(f"oracle forward (bf16, CPU): {time.time() - t0:.0f} s", flush=True)
t0 = time.time()
with torch.no_grad():   # the same restatement in fp32: the yardstick for what bf16 arithmetic itself costs at this depth
    sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    ref32 = O.forward_eval(sd32, cfg, pv.float(), toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion.float(),
                           toks["img_context_token_id"], mos=None, stage=2)
    del sd32
s32 = ref32["score1"].float().item()
print(f"oracle forward (fp32, CPU): {time.time() - t0:.0f} s; score1 fp32 {s32:.6f}, bf16 oracle {ref['score1'].float().item():.6f} "
      f"(|bf16 oracle - fp32| {abs(ref['score1'].float().item() - s32):.6f}); level tokens bf16 oracle vs fp32: "
      f"{int((ref['logit'][ref['label'] != -100] == ref32['logit'][ref32['label'] != -100]).sum())}/{int((ref['label'] != -100).sum())}", flush=True)
model = InternVLChatModel(cfg, device=torch.device("cuda", 0), max_clips=B, max_frames=B * T, max_tokens=B * toks["input_ids"].shape[1])
model.load_state_dict(sd)
model.img_context_token_id = toks["img_context_token_id"]
model.eval()
for name, trim in (("dead-row elimination on (default)", True), ("every row of every token (reference's amount of work)", False)):
    model.set_row_trimming(trim)
    out = model(mos=None, pixel_values=pv, input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags,
                labels=toks["labels"], motion_feature=motion)
    torch.cuda.synchronize()
    want = ref["label"] != -100
    got, exp = out["logit"].cpu()[want], ref["logit"][want]
    logits = ref["logits"][..., :-1, :].reshape(-1, ref["logits"].shape[-1])[want].float()
    flips = (got != exp).nonzero().flatten().tolist()
    gaps = []
    for r in flips:
        a, b = logits[r, exp[r]].item(), logits[r, got[r]].item()
        ulp = 2.0 ** (torch.tensor(abs(a)).clamp_min(1e-30).log2().floor().item() - 7)
        gaps.append(round(abs(a - b) / ulp, 2))
    s_hip, s_ref = out["score1"].float().cpu().item(), ref["score1"].float().item()
    print(f"{name}: |hip - fp32| {abs(s_hip - s32):.6f}; score1 hip {s_hip:.6f} oracle {s_ref:.6f} |d| {abs(s_hip - s_ref):.6f} (bf16 ulp at this value {2.0 ** (torch.tensor(abs(s_ref)).log2().floor().item() - 7):.6f}); "
          f"level tokens identical {int((got == exp).sum())}/{int(want.sum())}; oracle's own logit gap (bf16 ulps) between its token and ours on the others: {gaps}", flush=True)
