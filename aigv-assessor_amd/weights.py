"""Checkpoint surgery on the host: fold LoRA adapters into plain weights at load time so stage-2
LoRA checkpoints run on the fused kernels without peft.

The reference wraps the ViT / LLM with peft (modeling_internvl_chat.py:275-305: r, lora_alpha = 2r, targets
attn.qkv / attn.proj / mlp.fc1 / mlp.fc2 and wqkv / wo / w1 / w2 / w3), saves only the adapter tensors
(stage2_train.py:223-235 ``lora_weights.pth``) and merges with ``merge_and_unload`` (tools/merge_lora.py:19-26).
peft is absent here, so the merge restates its published rule  W' = W + (alpha / r) * B @ A  — parity unpinned.
"""
from __future__ import annotations

import re
from typing import Dict, Optional

import torch

_PEFT_WRAP = re.compile(r"\.base_model\.model\.")


def strip_peft_names(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``vision_model.base_model.model.encoder…`` -> ``vision_model.encoder…``; ``….base_layer.weight`` -> ``….weight``."""
    out = {}
    for k, v in sd.items():
        k2 = _PEFT_WRAP.sub(".", k).replace(".base_layer.", ".")
        out[k2] = v
    return out


def merge_lora_state_dict(base: Dict[str, torch.Tensor], lora: Optional[Dict[str, torch.Tensor]] = None,
                          alpha_over_r: float = 2.0, adapter: str = "default") -> Dict[str, torch.Tensor]:
    """Return a plain state-dict with every ``X.lora_A.<adapter>.weight`` [r, in] / ``X.lora_B.<adapter>.weight``
    [out, r] pair folded into ``X.weight``.  ``lora`` may be the separate ``lora_weights.pth`` dict; adapter tensors
    found inside ``base`` are merged too.  ``alpha_over_r`` = lora_alpha / r (the reference always uses 2)."""
    sd = strip_peft_names(base)
    extra = strip_peft_names(lora) if lora else {}
    pool = dict(sd)
    pool.update(extra)
    merged = {k: v for k, v in sd.items() if ".lora_A." not in k and ".lora_B." not in k}
    a_tag = f".lora_A.{adapter}.weight"
    for k, a in pool.items():
        if not k.endswith(a_tag):
            continue
        stem = k[: -len(a_tag)]
        b = pool.get(f"{stem}.lora_B.{adapter}.weight")
        w_key = stem + ".weight"
        if b is None or w_key not in merged:
            raise KeyError(f"incomplete LoRA triple for {stem}")
        w = merged[w_key]
        delta = (b.to(torch.float32) @ a.to(torch.float32)) * alpha_over_r
        merged[w_key] = (w.to(torch.float32) + delta.to(w.device)).to(w.dtype)
    return merged


# ---- the reference's second LLM family: transformers' LlamaForCausalLM (modeling_internvl_chat.py:228-229) ---------------------------
# The HIP path keeps ONE weight layout - InternLM2's (packed wqkv in (kv head, [q heads of the group | K | V], d) row order, w1 / w3 / w2):
# a Llama checkpoint is re-packed on the host at load time.  Every output feature of the packed matrices is one row of one of the
# original matrices, so the linears compute exactly the original dot products; RMSNorm, rotate-half RoPE (HF layout in both families),
# the eager attention and the SwiGLU MLP are the same functions in both (transformers' modeling_llama.py is where InternLM2's code came from).
_LLAMA_LAYER = re.compile(r"^language_model\.model\.layers\.(\d+)\.(.+)$")
_LLAMA_RENAMES = {
    "self_attn.o_proj.weight": "attention.wo.weight",
    "mlp.gate_proj.weight": "feed_forward.w1.weight",
    "mlp.up_proj.weight": "feed_forward.w3.weight",
    "mlp.down_proj.weight": "feed_forward.w2.weight",
    "input_layernorm.weight": "attention_norm.weight",
    "post_attention_layernorm.weight": "ffn_norm.weight",
}


def is_llama_state_dict(sd) -> bool:
    return "language_model.model.embed_tokens.weight" in sd or "language_model.lm_head.weight" in sd


def llama_to_internlm2(sd: Dict[str, torch.Tensor], llm_cfg) -> Dict[str, torch.Tensor]:
    """HF Llama names / separate q, k, v projections -> the InternLM2 names and packing the kernels read.  Tensors outside
    ``language_model.`` pass through.  Biases (``attention_bias`` / ``mlp_bias`` checkpoints) are refused: the reference's configs have none."""
    nh, nkv, d = llm_cfg.num_attention_heads, llm_cfg.num_key_value_heads, llm_cfg.head_dim
    g = nh // nkv
    out: Dict[str, torch.Tensor] = {}
    qkv: Dict[int, Dict[str, torch.Tensor]] = {}
    for k, v in sd.items():
        if k == "language_model.model.embed_tokens.weight":
            out["language_model.model.tok_embeddings.weight"] = v
            continue
        if k == "language_model.lm_head.weight":
            out["language_model.output.weight"] = v
            continue
        m = _LLAMA_LAYER.match(k)
        if not m:
            out[k] = v
            continue
        i, rest = int(m.group(1)), m.group(2)
        if rest in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight"):
            qkv.setdefault(i, {})[rest.split(".")[1][0]] = v
        elif rest in _LLAMA_RENAMES:
            out[f"language_model.model.layers.{i}.{_LLAMA_RENAMES[rest]}"] = v
        elif rest.endswith(".bias"):
            raise NotImplementedError(f"{k}: Llama checkpoints with projection biases are not on this path")
        elif rest == "self_attn.rotary_emb.inv_freq":
            continue    # a buffer older transformers versions saved; the tables are rebuilt from the config
        else:
            raise KeyError(f"unexpected Llama tensor {k}")
    for i, t in qkv.items():
        if set(t) != {"q", "k", "v"}:
            raise KeyError(f"layer {i}: incomplete q / k / v projections ({sorted(t)})")
        H = t["q"].shape[1]
        if tuple(t["q"].shape) != (nh * d, H) or tuple(t["k"].shape) != (nkv * d, H) or tuple(t["v"].shape) != (nkv * d, H):
            raise RuntimeError(f"layer {i}: q / k / v shapes {tuple(t['q'].shape)} {tuple(t['k'].shape)} {tuple(t['v'].shape)} do not match the config")
        packed = torch.cat([t["q"].view(nkv, g, d, H), t["k"].view(nkv, 1, d, H), t["v"].view(nkv, 1, d, H)], dim=1)
        out[f"language_model.model.layers.{i}.attention.wqkv.weight"] = packed.reshape((g + 2) * nkv * d, H)
    return out


def internlm2_to_llama(sd: Dict[str, torch.Tensor], llm_cfg) -> Dict[str, torch.Tensor]:
    """The inverse re-packing (tests and fixture generators: one seeded weight set serves both families)."""
    nh, nkv, d = llm_cfg.num_attention_heads, llm_cfg.num_key_value_heads, llm_cfg.head_dim
    g = nh // nkv
    inv = {v: k for k, v in _LLAMA_RENAMES.items()}
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k == "language_model.model.tok_embeddings.weight":
            out["language_model.model.embed_tokens.weight"] = v
            continue
        if k == "language_model.output.weight":
            out["language_model.lm_head.weight"] = v
            continue
        m = _LLAMA_LAYER.match(k)
        if not m:
            out[k] = v
            continue
        i, rest = int(m.group(1)), m.group(2)
        p = f"language_model.model.layers.{i}."
        if rest == "attention.wqkv.weight":
            H = v.shape[1]
            w = v.view(nkv, g + 2, d, H)
            out[p + "self_attn.q_proj.weight"] = w[:, :g].reshape(nh * d, H).clone()
            out[p + "self_attn.k_proj.weight"] = w[:, g].reshape(nkv * d, H).clone()
            out[p + "self_attn.v_proj.weight"] = w[:, g + 1].reshape(nkv * d, H).clone()
        elif rest in inv:
            out[p + inv[rest]] = v
        else:
            raise KeyError(f"unexpected InternLM2 tensor {k}")
    return out


def llama_stream_to_internlm2(named_tensors, llm_cfg):
    """llama_to_internlm2 for an iterable of (name, tensor) (InternVLChatModel.load_state_dict_stream): renames pass straight through, a
    layer's q / k / v projections are held until the third arrives and leave as one packed wqkv - at most one layer's projections are alive."""
    pending: Dict[int, Dict[str, torch.Tensor]] = {}
    for k, v in named_tensors:
        m = _LLAMA_LAYER.match(k)
        if k in ("language_model.model.tok_embeddings.weight", "language_model.output.weight") or (
                m and m.group(2).split(".")[0] in ("attention", "feed_forward", "attention_norm", "ffn_norm")):
            yield k, v      # already InternLM2 layout (what load_state_dict accepts too)
            continue
        if m and m.group(2) in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight"):
            i = int(m.group(1))
            pending.setdefault(i, {})[k] = v
            if len(pending[i]) == 3:
                yield from llama_to_internlm2(pending.pop(i), llm_cfg).items()
            continue
        yield from llama_to_internlm2({k: v}, llm_cfg).items()
    if pending:
        raise KeyError(f"incomplete q / k / v projections for layers {sorted(pending)}")
