"""Plain configuration structs for the scorer hot path.

Field names follow the reference's HF config classes so that a reference ``config.json`` loads
unchanged (reference: internvl/model/internvl_chat_eval2/configuration_intern_vit.py:20-119,
internvl/model/internlm2/configuration_internlm2.py:26-150,
internvl/model/internvl_chat_eval2/configuration_internvl_chat.py:20-108).  They are plain
dataclasses: the build does not subclass ``transformers.PretrainedConfig`` and
never opens the hard-coded ``/DATA/...`` path the reference opens (configuration_internvl_chat.py:42-46).

The widths the reference hard-codes in the model body (motion feature 2304, frame view 448,
score-head input 4096: modeling_internvl_chat.py:44,244-249,337) are parameters here with the
reference's values as defaults, so InternVL2-8B behaves identically.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field, fields
from typing import Optional, Tuple


def _pick(cls, d: dict):
    names = {f.name for f in fields(cls)}
    return cls(**{k: v for k, v in d.items() if k in names})


@dataclass
class InternVisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_attention_heads: int = 16
    num_hidden_layers: int = 24
    image_size: int = 448
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-6
    norm_type: str = "layer_norm"          # 'layer_norm' (ViT-300M) | 'rms_norm' (ViT-6B)
    qkv_bias: bool = True
    qk_normalization: bool = False
    hidden_act: str = "gelu"
    initializer_factor: float = 1.0
    initializer_range: float = 0.02
    use_flash_attn: bool = True

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def _normalise_rope_scaling(sc):
    """{'type': 'linear' | 'dynamic', 'factor': f} whatever spelling the configuration uses ('rope_type' is transformers' newer alias of
    'type'); 'default' / None -> None.  Anything else (llama3, yarn, longrope ...) is not built here: wrong RoPE is wrong logits with no
    error, so it raises instead of running unscaled."""
    if not sc:
        return None
    kind = sc.get("type", sc.get("rope_type"))
    if kind in (None, "default"):
        return None
    if kind not in ("linear", "dynamic"):
        raise NotImplementedError(f"rope scaling type {kind!r} is not on this path (linear and dynamic-NTK are)")
    if "factor" not in sc:
        raise ValueError(f"rope scaling {sc!r} has no factor")
    return {"type": kind, "factor": float(sc["factor"])}


@dataclass
class InternLM2Config:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    num_hidden_layers: int = 32
    vocab_size: int = 92553
    rms_norm_eps: float = 1e-5
    rope_theta: float = 1000000.0
    max_position_embeddings: int = 32768
    rope_scaling: Optional[dict] = None
    hidden_act: str = "silu"
    bias: bool = False
    initializer_range: float = 0.02
    pad_token_id: int = 2
    bos_token_id: int = 1
    eos_token_id: int = 2
    use_cache: bool = True
    attn_implementation: str = "flash_attention_2"
    architectures: Tuple[str, ...] = ("InternLM2ForCausalLM",)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class InternVLChatConfig:
    vision_config: InternVisionConfig = field(default_factory=InternVisionConfig)
    llm_config: InternLM2Config = field(default_factory=InternLM2Config)
    select_layer: int = -1
    force_image_size: Optional[int] = 448
    downsample_ratio: float = 0.5
    template: str = "internlm2-chat"
    ps_version: str = "v2"
    dynamic_image_size: bool = True
    use_thumbnail: bool = True
    min_dynamic_patch: int = 1
    max_dynamic_patch: int = 6
    use_backbone_lora: int = 0
    use_llm_lora: int = 0
    # widths the reference hard-codes (modeling_internvl_chat.py:44-51,244-249)
    motion_dim: int = 2304
    score_dims: Tuple[int, ...] = (1024, 256, 64, 16, 1)

    # ---- derived ----
    @property
    def image_size(self) -> int:
        return self.force_image_size or self.vision_config.image_size

    @property
    def grid(self) -> int:
        return self.image_size // self.vision_config.patch_size

    @property
    def num_image_token(self) -> int:
        # modeling_internvl_chat.py:211
        return int(self.grid ** 2 * (self.downsample_ratio ** 2))

    @property
    def proj_in(self) -> int:
        return self.vision_config.hidden_size * int(1 / self.downsample_ratio) ** 2

    @classmethod
    def from_dict(cls, d: dict) -> "InternVLChatConfig":
        d = dict(d)
        v = d.pop("vision_config", {}) or {}
        l = d.pop("llm_config", {}) or {}
        if isinstance(v, dict):
            v = _pick(InternVisionConfig, v)
        if isinstance(l, dict):
            l = dict(l)
            if "architectures" in l and l["architectures"] is not None:
                l["architectures"] = tuple(l["architectures"])
            if (l.get("architectures") or ("",))[0] == "LlamaForCausalLM":
                # transformers' LlamaConfig: the same field names as InternLM2Config (which was derived from it) with its OWN defaults
                # (configuration_llama.py); newer transformers versions nest the rotary base and scaling under rope_parameters
                rp = dict(l.get("rope_parameters") or {})
                l.setdefault("rope_theta", rp.get("rope_theta", 10000.0))
                for key, default in (("vocab_size", 32000), ("hidden_size", 4096), ("intermediate_size", 11008), ("num_hidden_layers", 32),
                                     ("num_attention_heads", 32), ("max_position_embeddings", 2048), ("rms_norm_eps", 1e-6),
                                     ("pad_token_id", None), ("bos_token_id", 1), ("eos_token_id", 2)):
                    l.setdefault(key, default)
                l.setdefault("num_key_value_heads", l["num_attention_heads"])
                if l.get("attention_bias") or l.get("mlp_bias"):
                    raise NotImplementedError("Llama configurations with projection biases are not on this path")
                if l.get("head_dim") not in (None, l["hidden_size"] // l["num_attention_heads"]):
                    raise NotImplementedError(f"head_dim {l['head_dim']} != hidden_size // num_attention_heads is not on this path")
                sc = l.get("rope_scaling")
                if sc is None and (rp.get("rope_type") or rp.get("type") or "default") != "default":
                    sc = rp                                                      # scaling carried inside rope_parameters
                l["rope_scaling"] = _normalise_rope_scaling(sc)
            elif l.get("rope_scaling") is not None:
                l["rope_scaling"] = _normalise_rope_scaling(l["rope_scaling"])
            l = _pick(InternLM2Config, l)
        cfg = _pick(cls, d)
        cfg.vision_config, cfg.llm_config = v, l
        if isinstance(cfg.score_dims, list):
            cfg.score_dims = tuple(cfg.score_dims)
        return cfg

    @classmethod
    def from_pretrained(cls, path: str, **overrides) -> "InternVLChatConfig":
        import os
        p = os.path.join(path, "config.json") if os.path.isdir(path) else path
        with open(p) as f:
            cfg = cls.from_dict(json.load(f))
        for k, v in overrides.items():
            setattr(cfg, k, v)
        return cfg

    def to_dict(self) -> dict:
        return asdict(self)


# ---------------------------------------------------------------------------------------------
# Named configurations (BASELINE.json `configs`; dims: E2/config.json for 8B, public model cards
# for 1B / 26B as recorded in SURVEY.md §8d).
# ---------------------------------------------------------------------------------------------
def internvl2_8b(**kw) -> InternVLChatConfig:
    """InternViT-300M + InternLM2.5-7B (internvl_chat_eval2/config.json:15-200)."""
    return InternVLChatConfig(**kw)


def internvl2_26b(**kw) -> InternVLChatConfig:
    """InternViT-6B (RMSNorm + QK-norm) + InternLM2-20B — model-card dims, not in the tree."""
    v = InternVisionConfig(hidden_size=3200, intermediate_size=12800, num_attention_heads=25,
                           num_hidden_layers=45, norm_type="rms_norm", qkv_bias=False,
                           qk_normalization=True)
    l = InternLM2Config(hidden_size=6144, intermediate_size=16384, num_attention_heads=48,
                        num_key_value_heads=8, num_hidden_layers=48, vocab_size=92553)
    return InternVLChatConfig(vision_config=v, llm_config=l, **kw)


def tiny(vit_hidden=128, vit_heads=2, vit_layers=2, vit_inter=256, llm_hidden=512, llm_heads=4,
         llm_kv_heads=2, llm_layers=2, llm_inter=768, vocab=1024, image_size=448,
         norm_type="layer_norm", qk_norm=False, qkv_bias=True, score_dims=(128, 64, 32, 16, 1),
         motion_dim=2304) -> InternVLChatConfig:
    """Small seeded configuration for parity tests (kernel-supported head dims 64 / 128)."""
    v = InternVisionConfig(hidden_size=vit_hidden, intermediate_size=vit_inter,
                           num_attention_heads=vit_heads, num_hidden_layers=vit_layers,
                           image_size=image_size, norm_type=norm_type, qk_normalization=qk_norm,
                           qkv_bias=qkv_bias)
    l = InternLM2Config(hidden_size=llm_hidden, intermediate_size=llm_inter,
                        num_attention_heads=llm_heads, num_key_value_heads=llm_kv_heads,
                        num_hidden_layers=llm_layers, vocab_size=vocab)
    return InternVLChatConfig(vision_config=v, llm_config=l, force_image_size=image_size,
                              score_dims=tuple(score_dims), motion_dim=motion_dim)
