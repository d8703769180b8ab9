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
