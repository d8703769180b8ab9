"""Golden vectors at BASELINE config 4's WIDTHS from the REFERENCE's own layer classes (round 5): one InternViT-6B encoder layer (hidden 3200,
25 heads of 128, RMSNorm + QK-norm, no qkv bias: internvl/model/internvl_chat_eval2/modeling_intern_vit.py:192-228) and one InternLM2-20B decoder
layer (hidden 6144, 48 query / 8 kv heads of 128, intermediate 16384: internvl/model/internlm2/modeling_internlm2.py:616-690), bf16 and fp32.

The reference cannot RUN the 26B model end to end (its score head is hard-wired to a 4096-wide LLM, modeling_internvl_chat.py:44,244-249), so the
full-depth config-4 fixture (make_golden_26b.py) is oracle-only; these two layers pin the oracle against the reference at exactly those widths
(N = 3200 = 12 x 256 + 128 and 9600 are the widths the GEMM dispatch splits by columns).  Neither weights nor inputs are stored: the test regenerates them from the recorded
seeds (synth.make_state_dict; torch.randn with a seeded generator); the outputs are kept at every fourth column.  Run (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_26b_layers.py

Output: tests/golden/layers_26b.pt
"""
import contextlib
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

SEED = 2605
VOCAB = 256          # (the layers under test do not touch the embeddings: a small table keeps the seeded state dict small)


def cfg_26b_one_layer():
    import aigv_assessor_amd as pkg
    cfg = pkg.internvl2_26b()
    cfg.vision_config.num_hidden_layers = 1
    cfg.llm_config.num_hidden_layers = 1
    cfg.llm_config.vocab_size = VOCAB
    return cfg


def main():
    import ref_shims
    from aigv_assessor_amd import synth
    from make_golden import E2E_LLM, E2E_VIS
    ref_shims.install(dict(E2E_LLM), dict(E2E_VIS))
    import internvl.model.internvl_chat_eval2.modeling_intern_vit as rvit
    import internvl.model.internlm2.modeling_internlm2 as rlm
    from internvl.model.internvl_chat_eval2.configuration_intern_vit import InternVisionConfig as RVisCfg
    from internvl.model.internlm2.configuration_internlm2 import InternLM2Config as RLmCfg

    cfg = cfg_26b_one_layer()
    v, l = cfg.vision_config, cfg.llm_config
    out = dict(seed=SEED, vocab=VOCAB, cases={})
    for dt in (torch.bfloat16, torch.float32):
        sd = synth.make_state_dict(cfg, seed=SEED, dtype=dt, rich=True)
        with contextlib.redirect_stdout(io.StringIO()):
            rc = RVisCfg(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_attention_heads=v.num_attention_heads, num_hidden_layers=1,
                         image_size=448, patch_size=14, norm_type="rms_norm", qk_normalization=True, qkv_bias=False, layer_norm_eps=v.layer_norm_eps,
                         drop_path_rate=0.0, use_flash_attn=False)
            layer = rvit.InternVisionEncoderLayer(rc, 0.0).to(dt).eval()
        pre = "vision_model.encoder.layers.0."
        layer.load_state_dict({k[len(pre):]: t for k, t in sd.items() if k.startswith(pre)}, strict=True)
        x = (torch.randn(2, 65, v.hidden_size, generator=torch.Generator().manual_seed(7)) * 0.5).to(dt)
        with torch.no_grad():
            y = layer(x)
        out["cases"][f"vit_layer/{dt}"] = dict(x_seed=7, x_shape=list(x.shape), x_scale=0.5, y_sub=y[..., ::4].contiguous().clone())
        del layer
        with contextlib.redirect_stdout(io.StringIO()):
            rl = RLmCfg(hidden_size=l.hidden_size, intermediate_size=l.intermediate_size, num_attention_heads=l.num_attention_heads,
                        num_key_value_heads=l.num_key_value_heads, num_hidden_layers=1, vocab_size=VOCAB, rms_norm_eps=l.rms_norm_eps, rope_theta=l.rope_theta,
                        max_position_embeddings=32768, rope_scaling={"factor": 2.0, "type": "dynamic"}, attn_implementation="eager", bias=False)
            rl.attn_implementation = "eager"
            dec = rlm.InternLM2DecoderLayer(rl).to(dt).eval()
        pre = "language_model.model.layers.0."
        dec.load_state_dict({k[len(pre):]: t for k, t in sd.items() if k.startswith(pre)}, strict=True)
        n = 40
        x = torch.randn(2, n, l.hidden_size, generator=torch.Generator().manual_seed(8)).to(dt)
        am = torch.ones(2, n, dtype=torch.bool)
        mask = rlm._expand_mask(am, dt, tgt_len=n) + rlm._make_causal_mask((2, n), dt, device=x.device)
        with torch.no_grad():
            y = dec(x, attention_mask=mask, position_ids=torch.arange(n).unsqueeze(0))[0]
        out["cases"][f"llm_layer/{dt}"] = dict(x_seed=8, x_shape=list(x.shape), x_scale=1.0, y_sub=y[..., ::4].contiguous().clone())
        del dec, sd
        print(dt, "vit layer out mean |y|", float(out["cases"][f"vit_layer/{dt}"]["y_sub"].float().abs().mean()), "llm layer out mean |y|",
              float(out["cases"][f"llm_layer/{dt}"]["y_sub"].float().abs().mean()), flush=True)
    dst = os.path.join(HERE, "layers_26b.pt")
    torch.save(out, dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KB")


if __name__ == "__main__":
    main()
