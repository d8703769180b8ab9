"""Diagnostic: how far are the HIP path and the oracle's bf16 path from the fp32 oracle on score1? (several seeds)"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from aigv_assessor_amd.modeling import InternVLChatModel
from oracle import oracle as O
BF = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "golden"
if which == "golden":
    e2e = torch.load("tests/golden/e2e.pt", weights_only=False)
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=e2e["vision_config"], llm_config=e2e["llm_config"], force_image_size=448, select_layer=-1))
    px, T = 448, 8
else:
    cfg = pkg.tiny(image_size=224, llm_layers=4, vit_layers=4)
    px, T = 224, 4
rows = []
for seed in range(21, 21 + int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    B = 1
    toks = synth.canonical_tokens(cfg, B, T, seed=seed)
    flags = torch.ones(B * T, 1, dtype=torch.long)
    res = {}
    for dt in (torch.float32, BF):
        sd = synth.make_state_dict(cfg, seed=seed, dtype=BF, rich=True)
        sdd = {k: v.to(dt) for k, v in sd.items()}
        pv = synth.synthetic_frames(B * T, px, seed=seed).to(dt)
        mo = synth.synthetic_motion(B, cfg.motion_dim, seed=seed).to(dt)
        r = O.forward_eval(sdd, cfg, pv, toks["input_ids"], toks["attention_mask"], flags, toks["labels"], mo, toks["img_context_token_id"], stage=2, return_intermediates=True)
        res[dt] = r
    model = InternVLChatModel(cfg); model.load_state_dict(sd); model.img_context_token_id = toks["img_context_token_id"]; model.eval().cuda()
    out = model(pixel_values=synth.synthetic_frames(B * T, px, seed=seed), input_ids=toks["input_ids"], attention_mask=toks["attention_mask"], image_flags=flags, labels=toks["labels"], motion_feature=synth.synthetic_motion(B, cfg.motion_dim, seed=seed))
    f32, b16, hip = res[torch.float32]["score1"].item(), res[BF]["score1"].float().item(), out["score1"].float().item()
    want = res[BF]["label"] != -100
    lv_ref = res[BF]["logit"][want]; lv_hip = out["logit"].cpu()[want]; lv_f32 = res[torch.float32]["logit"][want]
    # visual embeds
    ve = model.extract_feature(synth.synthetic_frames(B * T, px, seed=seed)).float().cpu().reshape(-1)
    vf = res[torch.float32]["vit_embeds"].float().reshape(-1); vb = res[BF]["vit_embeds"].float().reshape(-1)
    print(f"seed {seed}: fp32 {f32:.5f} bf16-oracle {b16:.5f} hip {hip:.5f} | |hip-f32| {abs(hip-f32):.5f} |bf16-f32| {abs(b16-f32):.5f} | levels hip!=bf16 {int((lv_ref!=lv_hip).sum())} bf16!=f32 {int((lv_ref!=lv_f32).sum())} hip!=f32 {int((lv_hip!=lv_f32).sum())}"
          f" | vis-emb rel err hip {((ve-vf).abs().mean()/vf.abs().mean()).item():.4f} bf16 {((vb-vf).abs().mean()/vf.abs().mean()).item():.4f}", flush=True)
    rows.append((abs(hip - f32), abs(b16 - f32), abs(hip - b16)))
    del model; torch.cuda.empty_cache()
import statistics as st
print("mean |hip-f32| %.5f  mean |bf16-f32| %.5f  mean |hip-bf16| %.5f  max |hip-bf16| %.5f" % (st.mean(r[0] for r in rows), st.mean(r[1] for r in rows), st.mean(r[2] for r in rows), max(r[2] for r in rows)))
