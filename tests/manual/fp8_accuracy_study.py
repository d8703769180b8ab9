"""Design input for BASELINE config 5 (fp8 weights), CPU only, no kernels: how far do the stage-2 outputs move when the four LLM
projections (wqkv, wo, w1|w3, w2) run on fp8 operands?  The oracle's F.linear is swapped, for language_model.model.layers.* only,
by a fake-quantised one: operands rounded to OCP e4m3 (torch.float8_e4m3fn, round-to-nearest-even) under a scaling recipe, product
accumulated in fp32, output rounded to bf16 like the real layer.  Recipes:
  token/channel : one scale per activation row and per weight row (amax / 448)   - unit MFMA scales, scaling in the epilogue
  mx32          : OCP MX blocks of 32 along K with power-of-two (E8M0) scales     - v_mfma_scale_* block scales
Synthetic random-init weights (the only ones available offline): this measures NUMERICAL drift of the forward pass, not task accuracy.
    PYTHONPATH=. python tests/manual/fp8_accuracy_study.py"""
import math, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import aigv_assessor_amd as pkg
from aigv_assessor_amd import synth
from oracle import oracle as O

E4M3_MAX = 448.0
def q_e4m3(x):  # fp32 -> e4m3 values (as fp32), RNE; inputs are pre-scaled into range
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)

def fq_rows(x):           # per-row amax scaling
    amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    s = amax / E4M3_MAX
    return q_e4m3(x / s) * s

def fq_mx32(x):           # MX: blocks of 32 along the last dim, power-of-two scale 2^(floor(log2 amax) - 8)  (e4m3 emax = 8)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -120)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    return (q_e4m3(xb / s) * s).reshape(shp)

MODE = {"token/channel": (fq_rows, fq_rows), "mx32": (fq_mx32, fq_mx32)}
real_linear = F.linear
def make_linear(mode):
    fa, fw = MODE[mode]
    def lin(x, w, b=None):
        if w.shape[0] in LLM_N and w.shape[1] in LLM_K and x.shape[-1] == w.shape[1] and ACTIVE[0]:
            y = real_linear(fa(x.float()), fw(w.float()))
            y = y if b is None else y + b.float()
            return y.to(x.dtype)
        return real_linear(x, w, b)
    return lin

ACTIVE = [False]
rows = []
CASES = (("tiny: 8 LLM layers, H 512", pkg.tiny(image_size=224, vit_layers=2, llm_layers=8), 2, 2),
         ("wide: 4 LLM layers, H 2048", pkg.tiny(image_size=224, vit_layers=1, llm_layers=4, llm_hidden=2048, llm_heads=16, llm_kv_heads=4,
                                                  llm_inter=5632, vocab=4096), 1, 2))
for (name, cfg, B, T) in CASES:
    l = cfg.llm_config
    LLM_N = {l.hidden_size, l.intermediate_size, (l.num_attention_heads + 2 * l.num_key_value_heads) * l.head_dim}
    LLM_K = {l.hidden_size, l.intermediate_size}
    for seed in (1, 2, 3):
        sd = synth.make_state_dict(cfg, seed=seed, rich=True)
        toks = synth.canonical_tokens(cfg, B, T, seed=seed)
        pv = synth.synthetic_frames(B * T, cfg.image_size, seed=seed)
        motion = synth.synthetic_motion(B, cfg.motion_dim, seed=seed)
        flags = torch.ones(B * T, 1, dtype=torch.long)
        def run(dtype):
            sdd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
            return O.forward_eval(sdd, cfg, pv.to(dtype), toks["input_ids"], toks["attention_mask"], flags, toks["labels"], motion.to(dtype),
                                  toks["img_context_token_id"], mos=None, stage=2)
        ACTIVE[0] = False
        ref32, ref16 = run(torch.float32), run(torch.bfloat16)
        want = ref16["label"] != -100
        res = {"bf16": ref16}
        for mode in MODE:
            F.linear = make_linear(mode); O.F.linear = F.linear
            ACTIVE[0] = True
            res[mode] = run(torch.bfloat16)
            ACTIVE[0] = False
            F.linear = real_linear; O.F.linear = real_linear
        for k, r in res.items():
            ds = (r["score1"].float() - ref32["score1"].float()).abs().max().item()
            flips = int((r["logit"][want] != ref32["logit"][want]).sum())
            rows.append((name, seed, k, ds, flips, int(want.sum())))
            print(f"{name:22s} seed {seed}  {k:14s} max|score1 - fp32 oracle| = {ds:.4f}   level tokens differing from fp32: {flips}/{int(want.sum())}", flush=True)
