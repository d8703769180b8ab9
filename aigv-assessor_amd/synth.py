"""Seeded synthetic weights and inputs (there are no checkpoints, tokenizer or videos offline).

State-dict names are the reference's (observed by instantiating
internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:195-273).  Initialisers follow the
reference where it defines them: Linear/Embedding ~ N(0, 0.02²) (internlm2/modeling_internlm2.py:728-737),
norm weights 1, biases 0, layer-scale = initializer_factor (modeling_intern_vit.py:211-212),
score head U(-0.15, 0.15) (modeling_internvl_chat.py:74).  ``rich=True`` perturbs every bias / norm
weight / layer-scale so that parity tests exercise those terms, and calibrates the last score layer so
``score1`` lands in the trained range (roughly 0.5 +- 0.4, still input-sensitive) instead of being
ReLU-clamped to 0 half of the time.

The canonical token layout is SURVEY.md §8d / Appendix A:  N = 73 + (7 + tokens_per_frame) * T.
"""
from __future__ import annotations

from typing import Optional, Dict, List, Tuple

import torch

from .config import InternVLChatConfig


def special_ids(vocab_size: int) -> Dict[str, int]:
    """<|im_end|>, <|im_start|>, <img>, </img>, <IMG_CONTEXT> = V-11 … V-7.

    For V = 92553 this gives 92542 … 92546, the reference's ids
    (conversation.py:381-385, stage2_eval.py:755-759)."""
    return {"im_end": vocab_size - 11, "im_start": vocab_size - 10, "img": vocab_size - 9,
            "img_end": vocab_size - 8, "img_context": vocab_size - 7}


def weight_shapes(cfg: InternVLChatConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in a fixed order. kind ∈ linear|embed|norm_w|bias|ls|cls|pos|score_w|score_b."""
    v, l = cfg.vision_config, cfg.llm_config
    Hv, Iv, P = v.hidden_size, v.intermediate_size, v.patch_size
    npos = (v.image_size // P) ** 2 + 1
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    e = "vision_model.embeddings."
    out += [(e + "class_embedding", (1, 1, Hv), "cls"),
            (e + "position_embedding", (1, npos, Hv), "pos"),
            (e + "patch_embedding.weight", (Hv, v.num_channels, P, P), "linear"),
            (e + "patch_embedding.bias", (Hv,), "bias")]
    for i in range(v.num_hidden_layers):
        p = f"vision_model.encoder.layers.{i}."
        out += [(p + "ls1", (Hv,), "ls"), (p + "ls2", (Hv,), "ls"),
                (p + "attn.qkv.weight", (3 * Hv, Hv), "linear")]
        if v.qkv_bias:
            out += [(p + "attn.qkv.bias", (3 * Hv,), "bias")]
        if v.qk_normalization:
            out += [(p + "attn.q_norm.weight", (Hv,), "norm_w"), (p + "attn.k_norm.weight", (Hv,), "norm_w")]
        out += [(p + "attn.proj.weight", (Hv, Hv), "linear"), (p + "attn.proj.bias", (Hv,), "bias"),
                (p + "mlp.fc1.weight", (Iv, Hv), "linear"), (p + "mlp.fc1.bias", (Iv,), "bias"),
                (p + "mlp.fc2.weight", (Hv, Iv), "linear"), (p + "mlp.fc2.bias", (Hv,), "bias"),
                (p + "norm1.weight", (Hv,), "norm_w"), (p + "norm2.weight", (Hv,), "norm_w")]
        if v.norm_type == "layer_norm":
            out += [(p + "norm1.bias", (Hv,), "bias"), (p + "norm2.bias", (Hv,), "bias")]
    H, I, V = l.hidden_size, l.intermediate_size, l.vocab_size
    d = l.head_dim
    qkv_out = (l.num_attention_heads + 2 * l.num_key_value_heads) * d
    out += [("language_model.model.tok_embeddings.weight", (V, H), "embed")]
    for i in range(l.num_hidden_layers):
        p = f"language_model.model.layers.{i}."
        out += [(p + "attention.wqkv.weight", (qkv_out, H), "linear"),
                (p + "attention.wo.weight", (H, l.num_attention_heads * d), "linear"),
                (p + "feed_forward.w1.weight", (I, H), "linear"),
                (p + "feed_forward.w3.weight", (I, H), "linear"),
                (p + "feed_forward.w2.weight", (H, I), "linear"),
                (p + "attention_norm.weight", (H,), "norm_w"),
                (p + "ffn_norm.weight", (H,), "norm_w")]
    out += [("language_model.model.norm.weight", (H,), "norm_w"),
            ("language_model.output.weight", (V, H), "linear")]
    Pin = cfg.proj_in
    out += [("mlp1.0.weight", (Pin,), "norm_w"), ("mlp1.0.bias", (Pin,), "bias"),
            ("mlp1.1.weight", (H, Pin), "linear"), ("mlp1.1.bias", (H,), "bias"),
            ("mlp1.3.weight", (H, H), "linear"), ("mlp1.3.bias", (H,), "bias")]
    M = cfg.motion_dim
    out += [("motion_mlp.0.weight", (M,), "norm_w"), ("motion_mlp.0.bias", (M,), "bias"),
            ("motion_mlp.1.weight", (H, M), "linear"), ("motion_mlp.1.bias", (H,), "bias"),
            ("motion_mlp.3.weight", (H, H), "linear"), ("motion_mlp.3.bias", (H,), "bias")]
    dims = (H,) + tuple(cfg.score_dims)
    for j in range(len(cfg.score_dims)):
        out += [(f"mlpscore.fc{j + 1}.weight", (dims[j + 1], dims[j]), "score_w"),
                (f"mlpscore.fc{j + 1}.bias", (dims[j + 1],), "score_b")]
    # mlpscore.ln1 exists in the reference state-dict but is unused (modeling_internvl_chat.py:55,85)
    out += [("mlpscore.ln1.weight", (cfg.score_dims[0],), "norm_w"), ("mlpscore.ln1.bias", (cfg.score_dims[0],), "bias")]
    return out


_M32 = 0xFFFFFFFF


def hashed_uniform(shape, key: int, std: float, dtype=torch.bfloat16, device="cpu", chunk: Optional[int] = None) -> torch.Tensor:
    """A seeded weight tensor that is the SAME BITS on every device: element e = ((h(e ^ key) >> 8) * 2^-24 - 0.5) * std * sqrt(12) with a
    32-bit integer hash h (two xorshift-multiply rounds with the key folded in before each, every intermediate < 2^63 in int64) - integer arithmetic, one exact int -> fp32
    conversion, two fp32 operations with exact or correctly rounded results, one rounding to ``dtype``.  Uniform with standard deviation
    ``std``.  Unlike ``torch.randn`` from a CPU generator (one serial mt19937 stream: ~4 ns per value on one core, 111 s for the 26 G
    parameters of InternVL2-26B) every element is a function of its own index, so a GPU fills 51 GB in seconds and a CPU can produce any
    tensor without walking the ones before it."""
    n = 1
    for d in shape:
        n *= int(d)
    if n >= 1 << 32:
        raise ValueError("hashed_uniform: tensors of 2^32 elements or more need a wider index hash")
    out = torch.empty(n, dtype=dtype, device=device)
    scale = torch.tensor(std * 12.0 ** 0.5, dtype=torch.float32, device=device)
    key2 = (int(key) * 2654435761 + 0x7F4A7C15) & _M32
    key = int(key) & _M32
    if chunk is None:      # (the values do not depend on it) CPU: cache-sized pieces, in place - 300 M values/s on 8 cores; GPU: few large launches
        chunk = 1 << 20 if torch.device(device).type == "cpu" else 1 << 26
    for e0 in range(0, n, chunk):
        x = torch.arange(e0, min(n, e0 + chunk), dtype=torch.int64, device=device)
        x.bitwise_xor_(key).bitwise_and_(_M32)
        for r in range(2):
            y = x >> 16
            y.bitwise_xor_(x).mul_(0x45D9F3B).bitwise_and_(_M32)
            if r == 0:
                y.bitwise_xor_(key2)      # (a key that enters before the first round only would make two tensors index permutations of each other)
            x = y
        y = x >> 16
        y.bitwise_xor_(x).bitwise_right_shift_(8)
        u = y.to(torch.float32)
        u.mul_(2.0 ** -24).sub_(0.5).mul_(scale)
        out[e0:e0 + chunk] = u.to(dtype)
    return out.view(*shape)


def make_state_dict_iter(cfg: InternVLChatConfig, seed: int = 0, dtype=torch.bfloat16, device="cpu", rich: bool = False, method: str = "randn",
                         big: bool = True):
    """(name, tensor) in weight_shapes order from ONE seeded generator - make_state_dict's values, one tensor alive at a time (a
    streaming consumer, tests/golden/make_golden_26b.py, walks 26 G parameters with a few hundred MB).

    ``method="hash"`` (round 6; the ORACLE-only InternVL2-26B fixture): Linear / Embedding matrices come from ``hashed_uniform`` (device
    independent bits; std 0.02 like the reference's ``_init_weights``), everything else (norms, biases, layer scales, position tables, score head:
    a few M values) from a CPU torch generator as before, then moved to ``device``.  ``big=False`` skips the matrices (the fixture generator's
    first pass wants the small tensors only).  The fixtures recorded from the imported REFERENCE all use ``method="randn"`` on the CPU."""
    if method == "hash":
        yield from _hashed_state_dict_iter(cfg, seed, dtype, device, rich, big)
        return
    if method != "randn":
        raise ValueError("method must be 'randn' or 'hash'")
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_score = len(cfg.score_dims)

    def randn(shape, std):
        return torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std

    def rand(shape, lo, hi):
        return torch.rand(shape, generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo

    for name, shape, kind in weight_shapes(cfg):
        if kind in ("linear", "embed"):
            if device != "cpu" and len(shape) == 2 and shape[0] * shape[1] > (1 << 26):
                # large matrices on the GPU: generate directly in the target dtype, row-chunked
                t = torch.empty(shape, dtype=dtype, device=device)
                step = max(1, (1 << 26) // shape[1])
                for r in range(0, shape[0], step):
                    t[r:r + step] = (torch.randn((min(step, shape[0] - r), shape[1]), generator=g,
                                                 device=device, dtype=torch.float32) * 0.02).to(dtype)
                yield name, t
                continue
            t = randn(shape, 0.02)
        elif kind in ("cls", "pos"):
            t = randn(shape, 0.02 if not rich else 0.1)
        elif kind == "norm_w":
            t = torch.ones(shape, device=device) if not rich else 1.0 + randn(shape, 0.1)
        elif kind == "bias":
            t = torch.zeros(shape, device=device) if not rich else randn(shape, 0.02)
        elif kind == "ls":
            t = torch.full(shape, cfg.vision_config.initializer_factor, device=device) if not rich \
                else rand(shape, 0.5, 1.5)
        elif kind == "score_w":
            t = rand(shape, -0.15, 0.15)
            if rich and name == f"mlpscore.fc{n_score}.weight":
                t = t * 0.3
        elif kind == "score_b":
            t = torch.zeros(shape, device=device)
            if rich:
                t = randn(shape, 0.02)
                if name == f"mlpscore.fc{n_score}.bias":
                    t = torch.full(shape, 0.5, device=device)
        else:
            raise AssertionError(kind)
        yield name, t.to(dtype)


def _hashed_state_dict_iter(cfg, seed, dtype, device, rich, big):
    # (the small tensors follow the same distributions as the randn method above, from a CPU generator of their own that no matrix draws from; kept apart from that
    # method's loop on purpose - ITS draw order is what every reference-recorded fixture depends on and must never change)
    g = torch.Generator().manual_seed(seed)
    n_score = len(cfg.score_dims)
    for i, (name, shape, kind) in enumerate(weight_shapes(cfg)):
        if kind in ("linear", "embed"):
            if big:
                yield name, hashed_uniform(shape, seed * 1000003 + i * 7919 + 12345, 0.02, dtype, device)
            continue
        if kind in ("cls", "pos"):
            t = torch.randn(shape, generator=g) * (0.02 if not rich else 0.1)
        elif kind == "norm_w":
            t = torch.ones(shape) if not rich else 1.0 + torch.randn(shape, generator=g) * 0.1
        elif kind == "bias":
            t = torch.zeros(shape) if not rich else torch.randn(shape, generator=g) * 0.02
        elif kind == "ls":
            t = torch.full(shape, cfg.vision_config.initializer_factor) if not rich else torch.rand(shape, generator=g) + 0.5
        elif kind == "score_w":
            t = torch.rand(shape, generator=g) * 0.3 - 0.15
            if rich and name == f"mlpscore.fc{n_score}.weight":
                t = t * 0.3
        elif kind == "score_b":
            t = torch.zeros(shape)
            if rich:
                t = torch.randn(shape, generator=g) * 0.02
                if name == f"mlpscore.fc{n_score}.bias":
                    t = torch.full(shape, 0.5)
        else:
            raise AssertionError(kind)
        yield name, t.to(dtype).to(device)


def make_state_dict(cfg: InternVLChatConfig, seed: int = 0, dtype=torch.bfloat16, device="cpu",
                    rich: bool = False, method: str = "randn") -> Dict[str, torch.Tensor]:
    return dict(make_state_dict_iter(cfg, seed=seed, dtype=dtype, device=device, rich=rich, method=method))


def condition_state_dict(sd: Dict[str, torch.Tensor], cfg: InternVLChatConfig) -> Dict[str, torch.Tensor]:
    """In place: make a seeded iid weight set look like a TRAINED checkpoint where that matters for how rounding noise travels through the
    depth: InternViT's layer scales ``ls1`` / ``ls2`` x 0.1 (trained InternViT checkpoints hold ~0.1; config.json's
    ``initializer_factor`` only sets the init) and every InternLM2 ``wo`` / ``w2`` x 1 / sqrt(2 L) (the depth-scaled residual-branch
    init of GPT-2-style training; 0.125 at L = 32: exact in bf16), so that each block ADDS a small update to the residual stream
    instead of replacing it.  Deterministic (fp32 multiply, one rounding back to the tensor's dtype)."""
    L = cfg.llm_config.num_hidden_layers
    down = 1.0 / (2.0 * L) ** 0.5
    for k, v in sd.items():
        if k.startswith("vision_model.encoder.layers.") and (k.endswith(".ls1") or k.endswith(".ls2")):
            sd[k] = (v.float() * 0.1).to(v.dtype)
        elif k.startswith("language_model.model.layers.") and (k.endswith("attention.wo.weight") or k.endswith("feed_forward.w2.weight")):
            sd[k] = (v.float() * down).to(v.dtype)
    return sd


def canonical_tokens(cfg: InternVLChatConfig, n_clips: int, n_frames: int, seed: int = 0,
                     answer_len: int = 9) -> Dict[str, torch.Tensor]:
    """input_ids / labels / attention_mask in the canonical layout of SURVEY.md §8d."""
    V = cfg.llm_config.vocab_size
    sp = special_ids(V)
    hi = max(4, min(V - 553, sp["im_end"]) if V > 2000 else sp["im_end"])
    g = torch.Generator().manual_seed(1000 + seed)

    def text(n):
        return torch.randint(3, hi, (n,), generator=g).tolist()

    ntok = cfg.num_image_token
    rows, labs = [], []
    for _ in range(n_clips):
        ids: List[int] = text(40)
        for _f in range(n_frames):
            ids += text(4) + [sp["img"]] + [sp["img_context"]] * ntok + [sp["img_end"]] + text(1)
        ids += text(4) + [sp["img"]] + [sp["img_context"]] + [sp["img_end"]]
        ids += text(16)
        n_prompt = len(ids)
        ans = text(answer_len) + [sp["im_end"]]
        ids += ans
        lab = [-100] * n_prompt + ans
        rows.append(ids)
        labs.append(lab)
    input_ids = torch.tensor(rows, dtype=torch.long)
    labels = torch.tensor(labs, dtype=torch.long)
    return {"input_ids": input_ids, "labels": labels,
            "attention_mask": torch.ones_like(input_ids, dtype=torch.bool),
            "img_context_token_id": sp["img_context"], "im_end_id": sp["im_end"]}


def perspective_prompts(base: Dict[str, torch.Tensor], n_prompts: int, seed: int = 0, question_lens=(16, 11, 21, 16),
                        answer_len: int = 9) -> List[Dict[str, torch.Tensor]]:
    """Variants of a canonical prompt that share everything up to the motion block and differ in the question / answer tokens
    behind it - the shape of the reference's four quality perspectives (`...Motion Feature:
    <img><IMG_CONTEXT></img>{question}<|im_end|><|im_start|>assistant\n{answer}<|im_end|>`).  Prompt 0 is `base` itself."""
    ids0 = base["input_ids"]
    n_clips, n0 = ids0.shape
    cut = n0 - 16 - (answer_len + 1)                 # first question token of the canonical layout
    g = torch.Generator().manual_seed(7000 + seed)
    hi = int(min(base["im_end_id"], ids0.max().item() + 1))
    out = [dict(base)]
    for p in range(1, n_prompts):
        q = int(question_lens[p % len(question_lens)])
        rows, labs = [], []
        for b in range(n_clips):
            question = torch.randint(3, max(4, hi), (q,), generator=g).tolist()
            ans = torch.randint(3, max(4, hi), (answer_len,), generator=g).tolist() + [int(base["im_end_id"])]
            rows.append(ids0[b, :cut].tolist() + question + ans)
            labs.append([-100] * (cut + q) + ans)
        ids = torch.tensor(rows, dtype=torch.long)
        out.append({"input_ids": ids, "labels": torch.tensor(labs, dtype=torch.long),
                    "attention_mask": torch.ones_like(ids, dtype=torch.bool),
                    "img_context_token_id": base["img_context_token_id"], "im_end_id": base["im_end_id"]})
    return out


def canonical_len(cfg: InternVLChatConfig, n_frames: int, answer_len: int = 9) -> int:
    return 40 + n_frames * (7 + cfg.num_image_token) + 7 + 16 + answer_len + 1


def synthetic_frames(n_frames_total: int, image_size: int, seed: int = 0, dtype=torch.bfloat16,
                     device="cpu") -> torch.Tensor:
    """pixel_values ~ N(0,1) clipped to ±2.5 (≈ ImageNet-normalised range), NCHW."""
    g = torch.Generator(device=device).manual_seed(1234 + seed)
    x = torch.randn((n_frames_total, 3, image_size, image_size), generator=g, device=device,
                    dtype=torch.float32).clamp_(-2.5, 2.5)
    return x.to(dtype)


def synthetic_motion(n_clips: int, motion_dim: int, seed: int = 0, dtype=torch.bfloat16,
                     device="cpu") -> torch.Tensor:
    """SlowFast feature stand-in (the SlowFast branch is an INPUT: SURVEY.md §2 row 6)."""
    g = torch.Generator(device=device).manual_seed(4321 + seed)
    return torch.rand((n_clips, motion_dim), generator=g, device=device, dtype=torch.float32).to(dtype)


# ---------------------------------------------------------------------------------------------------------
# SlowFast-R50 motion branch: synthetic weights under the reference's state-dict path
# ---------------------------------------------------------------------------------------------------------
SLOWFAST_PREFIX = "slowfast_model.feature_extraction."
SLOWFAST_DEPTHS = (3, 4, 6, 3)


def slowfast_conv_shapes() -> List[Tuple[str, str, Tuple[int, ...]]]:
    """(conv name, norm name, weight shape [Cout, Cin, kt, kh, kw]) for blocks 0..4 of pytorchvideo's ``slowfast_r50`` - the
    sub-modules the reference keeps (modeling_internvl_chat.py:164-173) - names relative to ``feature_extraction.``."""
    out: List[Tuple[str, str, Tuple[int, ...]]] = []
    out.append(("0.multipathway_blocks.0.conv", "0.multipathway_blocks.0.norm", (64, 3, 1, 7, 7)))
    out.append(("0.multipathway_blocks.1.conv", "0.multipathway_blocks.1.norm", (8, 3, 5, 7, 7)))
    out.append(("0.multipathway_fusion.conv_fast_to_slow", "0.multipathway_fusion.norm", (16, 8, 7, 1, 1)))
    cin = [64 + 16, 8]
    for stage in range(4):
        for path in (0, 1):
            inner = (8 if path else 64) << stage
            cout = 4 * inner
            kt = 3 if (path == 1 or stage >= 2) else 1
            for blk in range(SLOWFAST_DEPTHS[stage]):
                p = f"{stage + 1}.multipathway_blocks.{path}.res_blocks.{blk}."
                c = cin[path] if blk == 0 else cout
                if blk == 0:
                    out.append((p + "branch1_conv", p + "branch1_norm", (cout, c, 1, 1, 1)))
                out.append((p + "branch2.conv_a", p + "branch2.norm_a", (inner, c, kt, 1, 1)))
                out.append((p + "branch2.conv_b", p + "branch2.norm_b", (inner, inner, 1, 3, 3)))
                out.append((p + "branch2.conv_c", p + "branch2.norm_c", (cout, inner, 1, 1, 1)))
            cin[path] = cout
        if stage < 3:
            cf = 32 << stage
            out.append((f"{stage + 1}.multipathway_fusion.conv_fast_to_slow", f"{stage + 1}.multipathway_fusion.norm", (2 * cf, cf, 7, 1, 1)))
            cin[0] += 2 * cf
    return out


def slowfast_state_dict(seed: int = 0, prefix: str = SLOWFAST_PREFIX) -> Dict[str, torch.Tensor]:
    """fp32 tensors: He-initialised convs and perturbed BatchNorm statistics; the last norm of every branch is scaled down
    so that 16 residual blocks keep activations O(1) in bf16."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for conv, norm, shape in slowfast_conv_shapes():
        fan_in = shape[1] * shape[2] * shape[3] * shape[4]
        sd[prefix + conv + ".weight"] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        c = shape[0]
        gain = 0.25 if norm.endswith("norm_c") else 1.0
        sd[prefix + norm + ".weight"] = (0.75 + 0.5 * torch.rand(c, generator=g)) * gain
        sd[prefix + norm + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + norm + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + norm + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[prefix + norm + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    return sd
