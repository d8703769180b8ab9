"""Full-DEPTH golden vectors at BASELINE config 4's widths (round 4): InternViT-6B (H 3200, 45 layers, RMSNorm + QK-norm,
25 heads of 128) + InternLM2-20B (H 6144, 48 layers, 48 q / 8 kv heads) on ONE 16-frame clip (N = 4281), stage-1 flavour (quality-level
tokens; the reference's score head is hard-wired to a 4096-wide LLM, modeling_internvl_chat.py:44,244-249, so the reference ITSELF cannot
run this width and there is no stage-2 score here).  ORACLE-ONLY: the vectors come from oracle/oracle.py (bit-exact with the imported
reference at the widths it can run: tests/test_oracle_golden.py), in bf16 AND fp32.

The 26 G parameters (51 GB in bf16) do not fit beside their activations in the 62 GB build container, so the pass STREAMS: the seeded
weight stream of synth.make_state_dict_iter is walked twice - once for the small tensors (mlp1 / motion_mlp are generated after the LLM
layers and needed in front of them), once more block by block while the oracle's layer functions run - and at most one layer's weights are
alive at a time.  The bf16 and the fp32 pass advance together through each layer's weights (the fp32 pass uses the bf16 weights upcast: one
model, two precisions).

Round 6: the weights are synth's ``method="hash"`` set (Linear / Embedding matrices from a device-independent integer hash, uniform with
std 0.02; the small tensors from a CPU torch generator).  Until round 5 they came from ONE serial CPU generator, and the GPU test spent
111 of its 115 s (and of the driver's 900 s for the whole suite) drawing 26 G normal deviates on one host core; the hashed set is the same
bits on CPU and GPU, so the test fills the model on the device in seconds.  The fixture records ``w_method``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_26b.py        (15 min on 8 otherwise idle cores with the hashed weights; 40 min with the serial generator of rounds 4-5; < 12 GB)

Output: tests/golden/e2e_26b_full.pt (plain tensors: loads with weights_only=True)
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

W_SEED, IN_SEED, T = 2601, 26, 16
W_METHOD = "hash"


def WEIGHT(cfg, name):
    """One matrix of the hashed weight set by name (every element is a function of its index: no walk over the tensors before it)."""
    for i, (n, shape, kind) in enumerate(synth.weight_shapes(cfg)):
        if n == name:
            return synth.hashed_uniform(shape, W_SEED * 1000003 + i * 7919 + 12345, 0.02)
    raise KeyError(name)


def block_of(name):
    p = name.split(".")
    if name.startswith("vision_model.encoder.layers."):
        return "v%d" % int(p[3])
    if name.startswith("language_model.model.layers."):
        return "l%d" % int(p[3])
    return "misc:" + name


def main():
    cfg = pkg.internvl2_26b()
    if os.environ.get("AIGV_GOLDEN_DRY"):
        cfg.vision_config.num_hidden_layers, cfg.llm_config.num_hidden_layers = 2, 2
    v, l = cfg.vision_config, cfg.llm_config
    toks = synth.canonical_tokens(cfg, 1, T, seed=IN_SEED)
    pv = synth.synthetic_frames(T, 448, seed=IN_SEED)
    motion = synth.synthetic_motion(1, cfg.motion_dim, seed=IN_SEED)
    ids, labels = toks["input_ids"], toks["labels"]
    N = ids.shape[1]
    t0 = time.time()
    # ---- one walk of the generator, consumed block by block in GENERATION order; the forward needs mlp1 / motion_mlp (generated after the
    # LLM layers) in front of the LLM, so the walk is done twice: walk 1 keeps the small tensors, walk 2 streams the layers
    small = {}
    for name, t in synth.make_state_dict_iter(cfg, seed=W_SEED, rich=True, method=W_METHOD, big=False):      # (norms, biases, layer scales, tables)
        small[name] = t
    for name, shape, kind in synth.weight_shapes(cfg):                                                          # ... and the matrices outside the layers
        if kind in ("linear", "embed") and block_of(name).startswith("misc:") and name not in ("language_model.model.tok_embeddings.weight", "language_model.output.weight"):
            small[name] = WEIGHT(cfg, name)
    print(f"walk 1 (small tensors) {time.time() - t0:.0f} s", flush=True)

    def up(sd):
        return {k: (t.float() if t.is_floating_point() else t) for k, t in sd.items()}
    small32 = up(small)
    x16 = x32 = None
    h16 = h32 = None
    mask16 = mask32 = pos = None
    cur, cur_block = {}, None

    def run_block(block, sd):
        nonlocal x16, x32, h16, h32
        sd32 = up(sd)
        with torch.no_grad():
            if block[0] == "v":
                i = int(block[1:])
                x16 = O.vit_layer({**small, **sd}, cfg, i, x16)
                x32 = O.vit_layer({**small32, **sd32}, cfg, i, x32)
            else:
                i = int(block[1:])
                h16, _ = O.llm_layer(sd, cfg, i, h16, mask16, pos)
                h32, _ = O.llm_layer(sd32, cfg, i, h32, mask32, pos)
        print(f"  {block} done ({time.time() - t0:.0f} s)", flush=True)

    with torch.no_grad():
        x16 = O.vit_embeddings(small, cfg, pv)
        x32 = O.vit_embeddings(small32, cfg, pv.float())
    emb_w = None
    for name, t in synth.make_state_dict_iter(cfg, seed=W_SEED, rich=True, method=W_METHOD):
        b = block_of(name)
        if b != cur_block and cur:
            run_block(cur_block, cur)
            cur = {}
        cur_block = b
        if b.startswith("misc:"):
            if name == "language_model.model.tok_embeddings.weight":
                # the ViT is complete: projector, motion token, scatter -> LLM input (modeling_internvl_chat.py:320-400)
                with torch.no_grad():
                    sd_e = {**small, name: t}
                    vit16 = O.projector(small, "mlp1", O.shuffled_tokens(x16, cfg.downsample_ratio))
                    vit32 = O.projector(small32, "mlp1", O.shuffled_tokens(x32, cfg.downsample_ratio))
                    mo16 = O.projector(small, "motion_mlp", motion.view(1, -1))
                    mo32 = O.projector(small32, "motion_mlp", motion.float().view(1, -1))
                    h16 = O.scatter_embeds(sd_e, ids, toks["img_context_token_id"], vit16, mo16)
                    h32 = O.scatter_embeds(up(sd_e), ids, toks["img_context_token_id"], vit32, mo32)
                    am = torch.ones(1, N, dtype=torch.bool)
                    mask16, mask32 = O.additive_mask(am, N, 0, torch.bfloat16), O.additive_mask(am, N, 0, torch.float32)
                    pos = torch.arange(N).unsqueeze(0)
                del sd_e
                print(f"LLM input ready ({time.time() - t0:.0f} s)", flush=True)
            elif name == "language_model.output.weight":
                emb_w = t
            cur_block = None
            continue
        cur[name] = t
    if cur:
        run_block(cur_block, cur)
    with torch.no_grad():
        n16 = O.rms_norm_cast_then_scale(h16, small["language_model.model.norm.weight"], l.rms_norm_eps)
        n32 = O.rms_norm_cast_then_scale(h32, small32["language_model.model.norm.weight"], l.rms_norm_eps)
        want = (labels[:, 1:] != -100).reshape(-1)
        rows = want.nonzero().flatten()
        recs = {}
        for tag, n, w in (("bf16", n16, emb_w), ("fp32", n32, emb_w.float())):
            lg = torch.nn.functional.linear(n[0, :-1][rows], w).float()                     # lm-head on the answer rows (LM:1094-1096, CHAT1:331-366)
            tv, ti = lg.topk(4, dim=-1)
            recs[tag] = dict(logit=lg.argmax(-1).clone(), top_values=tv.clone(), top_ids=ti.clone(), hidden_m4=n[:, -4, :].clone(), row_sigma=lg.std(-1).clone())
            print(f"{tag}: answer-row argmax {recs[tag]['logit'].tolist()}", flush=True)
    out = dict(config="internvl2_26b", vit_layers=v.num_hidden_layers, llm_layers=l.num_hidden_layers, w_seed=W_SEED, w_method=W_METHOD, in_seed=IN_SEED, T=T, n_tokens=N,
               answer_rows=rows.clone(), label=labels[0, 1:][rows].clone(), cases=recs, seconds=time.time() - t0)
    path = os.path.join(HERE, "e2e_26b_full.pt") if not os.environ.get("AIGV_GOLDEN_DRY") else "/tmp/e2e_26b_dry.pt"
    torch.save(out, path)
    h = recs["bf16"]["hidden_m4"].float()
    print(f"wrote {path}; hidden[:, -4] rel L2 bf16 vs fp32 {float((h - recs['fp32']['hidden_m4']).norm() / recs['fp32']['hidden_m4'].norm()):.4f}; "
          f"levels bf16 vs fp32 agree on {int((recs['bf16']['logit'] == recs['fp32']['logit']).sum())}/{rows.numel()} ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
