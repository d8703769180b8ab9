"""Which share of |HIP - reference| at full depth is the attention kernel's own numerics?  (round 4, CPU only, ~25 min, ~45 GB)

The prefill attention kernel (csrc/attention.hip) is flash-style: scores stay fp32, P is rounded to bf16 UN-normalised and the row is
divided by its fp32 sum at the end.  The reference's eager path (modeling_internlm2.py:417-424, modeling_intern_vit.py:153-158) rounds
the score matrix to bf16 (LLM: twice - after q k^T and after / sqrt(d)), normalises in fp32 and rounds the NORMALISED P to bf16.  Both
are bf16-level noise, but noise the kernel does not share with the reference.  This script runs the ORACLE (bit-exact with the reference:
tests/test_oracle_golden.py) on the benched batch with the seeded full-size weights under the four combinations

    round_s (scores rounded as the reference does) x norm_p (P normalised before its bf16 rounding)

and prints each variant's distance to the reference's RECORDED bf16 / fp32 scores (tests/golden/e2e_8b_r3.pt): (True, True) must
reproduce the recording; (False, False) is the arithmetic of the shipped kernel with CPU summation order.

    PYTHONDONTWRITEBYTECODE=1 python tests/manual/attention_numerics_study.py [--clips 4] [--layers 32]
"""
import argparse
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

ROUND_S, NORM_P = True, True


def _softmax_pv(s32, v, dtype):
    """s32: fp32 scores (already masked with -inf / finfo.min), v: [.., keys, d] -> attention output in `dtype`."""
    if NORM_P:
        w = F.softmax(s32, dim=-1, dtype=torch.float32).to(dtype)
        return torch.matmul(w, v)
    m = s32.max(dim=-1, keepdim=True).values
    p = torch.exp(s32 - m)
    l = p.sum(dim=-1, keepdim=True)
    o = torch.matmul(p.to(dtype).float(), v.float()) / l
    return o.to(dtype)


def vit_attention(sd, cfg, prefix, x):
    v = cfg.vision_config
    nf, n, c = x.shape
    nh = v.num_attention_heads
    d = c // nh
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(nf, n, 3, nh, d).permute(2, 0, 3, 1, 4)
    q, k, vv = qkv[0], qkv[1], qkv[2]
    assert not v.qk_normalization
    qs = q * (d ** -0.5)
    if ROUND_S:
        att = (qs @ k.transpose(-2, -1)).float()           # the bf16 matmul output, i.e. the rounded scores
    else:
        att = qs.float() @ k.float().transpose(-2, -1)
    y = _softmax_pv(att, vv, x.dtype).transpose(1, 2).reshape(nf, n, c)
    return F.linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def llm_attention(sd, cfg, i, x, mask, position_ids, past=None):
    l = cfg.llm_config
    p = f"language_model.model.layers.{i}.attention."
    b, n, _ = x.shape
    nh, nkv, d = l.num_attention_heads, l.num_key_value_heads, l.head_dim
    g = nh // nkv
    qkv = F.linear(x, sd[p + "wqkv.weight"]).view(b, n, nkv, g + 2, d)
    q = qkv[..., :g, :].reshape(b, n, nh, d).transpose(1, 2)
    k = qkv[..., g, :].transpose(1, 2)
    v = qkv[..., g + 1, :].transpose(1, 2)
    cos, sin = O.rope_tables(d, l.rope_theta, n, v.dtype, l.max_position_embeddings, l.rope_scaling)
    q, k = O.apply_rope(q, k, cos, sin, position_ids)
    kr = k[:, :, None].expand(b, nkv, g, n, d).reshape(b, nh, n, d)
    vr = v[:, :, None].expand(b, nkv, g, n, d).reshape(b, nh, n, d)
    out = torch.empty(b, nh, n, d, dtype=x.dtype)
    for h0 in range(0, nh, 8):                           # head groups: bounds the fp32 score matrix
        sl = slice(h0, h0 + 8)
        if ROUND_S:
            w = (torch.matmul(q[:, sl], kr[:, sl].transpose(2, 3)) / math.sqrt(d) + mask).float()
        else:
            w = torch.matmul(q[:, sl].float(), kr[:, sl].float().transpose(2, 3)) / math.sqrt(d) + mask.float()
        out[:, sl] = _softmax_pv(w, vr[:, sl], x.dtype)
    y = out.transpose(1, 2).contiguous().reshape(b, n, nh * d)
    return F.linear(y, sd[p + "wo.weight"]), (k, v)


def main():
    global ROUND_S, NORM_P
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--variants", default="ff,tf,ft")
    args = ap.parse_args()
    g = torch.load(os.path.join(ROOT, "tests", "golden", "e2e_8b_r3.pt"), weights_only=True)
    r16, r32 = g["cases"]["batch4/bf16"], g["cases"]["batch4/fp32"]
    cfg = pkg.InternVLChatConfig.from_dict(dict(vision_config=g["vision_config"], llm_config=g["llm_config"], force_image_size=448, select_layer=-1))
    t0 = time.time()
    sd = synth.make_state_dict(cfg, seed=g["w_seed"], rich=True)
    for k, v in g["overrides"].items():
        sd[k] = torch.full_like(sd[k], v)
    sd = {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in sd.items()}
    print(f"weights in {time.time() - t0:.0f} s", flush=True)
    B, T, seed = args.clips, r16["T"], r16["seed"]
    toks = synth.canonical_tokens(cfg, r16["B"], T, seed=seed)
    pv = synth.synthetic_frames(r16["B"] * T, 448, seed=seed, dtype=torch.bfloat16)[: B * T]
    motion = synth.synthetic_motion(r16["B"], 2304, seed=seed, dtype=torch.bfloat16)[:B]
    O.vit_attention, O.llm_attention = vit_attention, llm_attention
    b16, f32 = r16["score1"].float()[:B], r32["score1"].float()[:B]
    ulp = 2.0 ** -8
    print(f"reference bf16 {b16.tolist()} fp32 {[round(x, 4) for x in f32.tolist()]}; |bf16 - fp32| mean {(b16 - f32).abs().mean():.4f}", flush=True)
    for name in args.variants.split(","):
        ROUND_S, NORM_P = name[0] == "t", name[1] == "t"
        t0 = time.time()
        with torch.no_grad():
            out = O.forward_eval(sd, cfg, pv, toks["input_ids"][:B], toks["attention_mask"][:B], torch.ones(B * T, 1, dtype=torch.long), toks["labels"][:B], motion,
                                 toks["img_context_token_id"], mos=torch.full((B,), 0.5, dtype=torch.bfloat16), stage=2)
        s = out["score1"].float()
        want = out["label"] != -100
        lev = out["logit"][want]
        n_lev = int((lev != r16["logit"][: lev.numel()]).sum())
        print(f"round_s={ROUND_S} norm_p={NORM_P}: score {s.tolist()} |d vs ref bf16| {[round(x / ulp, 1) for x in (s - b16).abs().tolist()]} ulps, mean "
              f"{(s - b16).abs().mean():.4f}; vs ref fp32 mean {(s - f32).abs().mean():.4f}; level tokens differing from ref bf16 {n_lev}/{lev.numel()} "
              f"({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
