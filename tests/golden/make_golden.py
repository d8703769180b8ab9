"""Generate golden vectors by running the REFERENCE's own CPU path in this container.

Run (build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules are imported through the four shims of ``ref_shims.py``.
Weights are not stored: every case records the seed and configuration, and the test regenerates the
identical weights with ``aigv_assessor_amd.synth.make_state_dict`` (CPU generator, deterministic).
Fixtures are data only — inputs (or their seeds) and the reference's outputs.

Outputs: tests/golden/components.pt, tests/golden/e2e.pt
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

import ref_shims  # noqa: E402
import aigv_assessor_amd as pkg  # noqa: E402
from aigv_assessor_amd import synth  # noqa: E402

E2E_LLM = dict(architectures=["InternLM2ForCausalLM"], hidden_size=4096, intermediate_size=512,
               num_attention_heads=32, num_key_value_heads=8, num_hidden_layers=2, vocab_size=640,
               rms_norm_eps=1e-5, rope_theta=1000000, max_position_embeddings=32768,
               rope_scaling={"factor": 2.0, "type": "dynamic"}, bias=False, hidden_act="silu",
               attn_implementation="eager", pad_token_id=2)
E2E_VIS = dict(architectures=["InternVisionModel"], hidden_size=128, intermediate_size=256,
               num_attention_heads=2, num_hidden_layers=2, image_size=448, patch_size=14,
               layer_norm_eps=1e-6, norm_type="layer_norm", qkv_bias=True, qk_normalization=False,
               hidden_act="gelu", drop_path_rate=0.0, initializer_factor=1.0, use_flash_attn=False)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def e2e_cfg():
    return pkg.InternVLChatConfig.from_dict(dict(vision_config=E2E_VIS, llm_config=E2E_LLM,
                                                 force_image_size=448, select_layer=-1))


def sub(t, rows=7, cols=5):
    """Strided subsample of the last two dims to keep fixtures small."""
    return t[..., ::rows, ::cols].contiguous().clone()


def main():
    m2, m1, cfg2, SlowFastStandIn = ref_shims.install(E2E_LLM, E2E_VIS)
    import internvl.model.internvl_chat_eval2.modeling_intern_vit as rvit
    import internvl.model.internlm2.modeling_internlm2 as rlm
    from internvl.model.internvl_chat_eval2.configuration_intern_vit import InternVisionConfig as RVisCfg
    from internvl.model.internlm2.configuration_internlm2 import InternLM2Config as RLmCfg

    torch.manual_seed(0)
    comp = {}

    # ---------------- components ----------------
    for flavour, norm_type, qkn, qkv_bias in (("ln", "layer_norm", False, True), ("rms", "rms_norm", True, False)):
        for dt in (torch.float32, torch.bfloat16):
            cfg = pkg.tiny(vit_hidden=128, vit_heads=2, vit_layers=1, vit_inter=256, norm_type=norm_type,
                           qk_norm=qkn, qkv_bias=qkv_bias)
            sd = synth.make_state_dict(cfg, seed=11, dtype=dt, rich=True)
            rc = RVisCfg(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_hidden_layers=1,
                         image_size=448, patch_size=14, norm_type=norm_type, qk_normalization=qkn,
                         qkv_bias=qkv_bias, layer_norm_eps=1e-6, drop_path_rate=0.0, use_flash_attn=False)
            layer = rvit.InternVisionEncoderLayer(rc, 0.0).to(dt).eval()
            pre = "vision_model.encoder.layers.0."
            layer.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
            x = (torch.randn(2, 65, 128, generator=torch.Generator().manual_seed(5)) * 0.5).to(dt)
            with torch.no_grad():
                y = layer(x)
            comp[f"vit_layer/{flavour}/{dt}"] = dict(seed=11, x=x, y=y)

    # ViT embeddings at 224 px on a 448-px position table (real bicubic 32x32 -> 16x16) and at 448
    for dt in (torch.float32, torch.bfloat16):
        for px in (224, 448):
            cfg = pkg.tiny(vit_hidden=64, vit_heads=1, vit_layers=1, vit_inter=128)
            sd = synth.make_state_dict(cfg, seed=12, dtype=dt, rich=True)
            rc = RVisCfg(hidden_size=64, intermediate_size=128, num_attention_heads=1, num_hidden_layers=1,
                         image_size=448, patch_size=14, use_flash_attn=False)
            emb = rvit.InternVisionEmbeddings(rc).to(dt).eval()
            pre = "vision_model.embeddings."
            emb.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
            pv = synth.synthetic_frames(2, px, seed=3, dtype=dt)
            with torch.no_grad():
                y = emb(pv)
            comp[f"vit_embed/{px}/{dt}"] = dict(seed=12, y=y)

    # InternLM2 decoder layer, eager attention, causal mask, d=128
    for dt in (torch.float32, torch.bfloat16):
        cfg = pkg.tiny(llm_hidden=256, llm_heads=2, llm_kv_heads=1, llm_layers=1, llm_inter=384, vocab=64)
        sd = synth.make_state_dict(cfg, seed=13, dtype=dt, rich=True)
        rc = RLmCfg(hidden_size=256, intermediate_size=384, num_attention_heads=2, num_key_value_heads=1,
                    num_hidden_layers=1, vocab_size=64, rms_norm_eps=1e-5, rope_theta=1000000,
                    max_position_embeddings=32768, rope_scaling={"factor": 2.0, "type": "dynamic"},
                    attn_implementation="eager", bias=False)
        rc.attn_implementation = "eager"
        layer = rlm.InternLM2DecoderLayer(rc).to(dt).eval()
        pre = "language_model.model.layers.0."
        layer.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
        n = 48
        x = (torch.randn(2, n, 256, generator=torch.Generator().manual_seed(6))).to(dt)
        am = torch.ones(2, n, dtype=torch.bool)
        mask = rlm._expand_mask(am, dt, tgt_len=n) + rlm._make_causal_mask((2, n), dt, device=x.device)
        pos = torch.arange(n).unsqueeze(0)
        with torch.no_grad():
            y = layer(x, attention_mask=mask, position_ids=pos)[0]
            rot = layer.attention.rotary_emb
            cos, sin = rot(x, seq_len=n)
        comp[f"llm_layer/{dt}"] = dict(seed=13, x=x, y=y, cos=cos.clone(), sin=sin.clone())

    # pixel shuffle v2 on a 6x6 grid (reference method needs only ps_version on self)
    class _PS:
        ps_version = "v2"
    x = torch.arange(2 * 6 * 6 * 8, dtype=torch.float32).reshape(2, 6, 6, 8)
    comp["pixel_shuffle"] = dict(x=x, y=m2.InternVLChatModel.pixel_shuffle(_PS(), x, scale_factor=0.5))

    torch.save(comp, os.path.join(HERE, "components.pt"))
    print("components:", len(comp), "cases")

    # ---------------- end to end ----------------
    cfg = e2e_cfg()
    e2e = {"llm_config": E2E_LLM, "vision_config": E2E_VIS}
    for tag, dt, B, T, seed in (("bf16_b1", torch.bfloat16, 1, 8, 21), ("fp32_b1", torch.float32, 1, 8, 21),
                                ("bf16_b2", torch.bfloat16, 2, 8, 22)):
        sd = synth.make_state_dict(cfg, seed=seed, dtype=dt, rich=True)
        with quiet():
            rcfg = cfg2.InternVLChatConfig(select_layer=-1, force_image_size=448, downsample_ratio=0.5,
                                           template="internlm2-chat", ps_version="v2")
            model = m2.InternVLChatModel(rcfg).to(dt).eval()
        missing = model.load_state_dict(sd, strict=True)
        toks = synth.canonical_tokens(cfg, B, T, seed=seed)
        model.img_context_token_id = toks["img_context_token_id"]
        pv = synth.synthetic_frames(B * T, 448, seed=seed, dtype=dt)
        motion = synth.synthetic_motion(B, 2304, seed=seed, dtype=dt)
        SlowFastStandIn.feature = motion
        mos = torch.full((B,), 0.5, dtype=dt)
        grabbed = {}
        hooks = [
            model.vision_model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("vit", o.last_hidden_state)),
            model.mlp1.register_forward_hook(lambda m, i, o: grabbed.__setitem__("mlp1", o)),
            model.motion_mlp.register_forward_hook(lambda m, i, o: grabbed.__setitem__("motion", o)),
            model.language_model.model.register_forward_hook(
                lambda m, i, o: grabbed.__setitem__("hidden", o.last_hidden_state)),
            model.language_model.model.layers[0].register_forward_hook(
                lambda m, i, o: grabbed.__setitem__("layer0", o[0])),
        ]
        with torch.no_grad(), quiet():
            out = model(mos=mos, pixel_values=pv, input_ids=toks["input_ids"],
                        attention_mask=toks["attention_mask"], image_flags=torch.ones(B * T, 1, dtype=torch.long),
                        labels=toks["labels"])
        for h in hooks:
            h.remove()
        rec = dict(seed=seed, B=B, T=T, dtype=str(dt), score1=out["score1"].clone(), loss=out["loss"].clone(),
                   label=out["label"].clone(), logit=out["logit"].clone(),
                   vit_sub=sub(grabbed["vit"]), mlp1_sub=sub(grabbed["mlp1"], 5, 37),
                   motion=grabbed["motion"].clone(), hidden_m4=grabbed["hidden"][:, -4, :].clone(),
                   hidden_sub=sub(grabbed["hidden"], 13, 41), layer0_sub=sub(grabbed["layer0"], 13, 41))
        # greedy decode through the reference's own cache path (manual loop; HF generate is gone in 5.x)
        if tag == "bf16_b1":
            lm = model.language_model
            with torch.no_grad(), quiet():
                n_prompt = int((toks["labels"][0] == -100).sum())
                ids = toks["input_ids"][:, :n_prompt]
                vit_embeds = model.extract_feature(pv)
                emb = lm.get_input_embeddings()(ids).clone()
                sel = ids.reshape(-1) == model.img_context_token_id
                csum = torch.cumsum(sel, 0)
                # generate(): all <IMG_CONTEXT> slots take visual tokens (CHAT:793-795); the prompt has
                # 8*256+1 slots, so feed the motion slot with the first visual token again (stage-1 style
                # prompts have no motion slot; this keeps the counts consistent for the decode check).
                flat = emb.reshape(-1, emb.shape[-1])
                src = torch.cat([vit_embeds.reshape(-1, emb.shape[-1]), vit_embeds.reshape(-1, emb.shape[-1])[:1]])
                flat[sel] = src
                emb = flat.reshape(1, -1, emb.shape[-1])
                am = torch.ones(1, n_prompt, dtype=torch.long)
                tokens = []
                o = lm(inputs_embeds=emb, attention_mask=am, position_ids=(am.cumsum(-1) - 1), use_cache=True)
                past = o.past_key_values
                for _ in range(6):
                    nxt = o.logits[:, -1, :].argmax(-1)
                    tokens.append(nxt)
                    am = torch.cat([am, torch.ones(1, 1, dtype=torch.long)], 1)
                    mi = lm.prepare_inputs_for_generation(nxt[:, None], past_key_values=past, attention_mask=am,
                                                          use_cache=True)
                    o = lm(**mi)
                    past = o.past_key_values
            rec["greedy_prompt_len"] = n_prompt
            rec["greedy_tokens"] = torch.stack(tokens, 1)
        e2e[tag] = rec
        print(tag, "score1", out["score1"].float().tolist(), "answer argmax", out["logit"][-11:-1].tolist())
        del model

    # stage-1 flavour (eval1 model: same pass with the motion token, returns {'label','logit'} only)
    sd = synth.make_state_dict(cfg, seed=23, dtype=torch.bfloat16, rich=True)
    with quiet():
        rcfg = cfg2.InternVLChatConfig(select_layer=-1, force_image_size=448, downsample_ratio=0.5,
                                       template="internlm2-chat", ps_version="v2")
        model = m1.InternVLChatModel(rcfg).to(torch.bfloat16).eval()
    keys = set(model.state_dict().keys())
    model.load_state_dict({k: v for k, v in sd.items() if k in keys}, strict=True)
    toks = synth.canonical_tokens(cfg, 1, 8, seed=23)
    model.img_context_token_id = toks["img_context_token_id"]
    pv = synth.synthetic_frames(8, 448, seed=23)
    SlowFastStandIn.feature = synth.synthetic_motion(1, 2304, seed=23)
    with torch.no_grad(), quiet():
        out = model(mos=torch.full((1,), 0.5, dtype=torch.bfloat16), pixel_values=pv, input_ids=toks["input_ids"],
                    attention_mask=toks["attention_mask"], image_flags=torch.ones(8, 1, dtype=torch.long),
                    labels=toks["labels"])
    e2e["stage1_bf16_b1"] = dict(seed=23, B=1, T=8, label=out["label"].clone(), logit=out["logit"].clone())
    print("stage1 answer argmax", out["logit"][-11:-1].tolist())

    torch.save(e2e, os.path.join(HERE, "e2e.pt"))
    print("wrote", os.path.join(HERE, "e2e.pt"))


if __name__ == "__main__":
    main()
