"""CPU ORACLE for the AIGV-Assessor scoring hot path  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path (``aigv-assessor_amd``)
never routes through it and has no CPU fallback.

What it is: a functional, op-for-op restatement in plain torch CPU ops of the reference's *CPU eager*
path (``flash_attn`` absent -> ``_naive_attn`` / eager ``InternLM2Attention``), with every rounding
point of SURVEY.md §8a-notes reproduced because it uses the same torch ops in the same dtype flow.
Every function cites the reference file:line it follows (paths relative to /root/reference/):

    VIT   = internvl/model/internvl_chat_eval2/modeling_intern_vit.py
    CHAT  = internvl/model/internvl_chat_eval2/modeling_internvl_chat.py   (stage-2 flavour)
    CHAT1 = internvl/model/internvl_chat_eval1/modeling_internvl_chat.py   (stage-1 flavour)
    LM    = internvl/model/internlm2/modeling_internlm2.py

Pinning: the reference ships no tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself, imported in the build container by
``tests/golden/make_golden.py`` (four shims) and committed under ``tests/golden/*.pt``;
``tests/test_oracle_golden.py`` checks every function here against them (bit-exact in that container).
The SlowFast motion branch is an INPUT on both sides (third-party, weights need a download:
parity unpinned for that branch, CHAT:135-193).

Weights are a flat ``dict[str, Tensor]`` with the reference's state-dict names;
``cfg`` is any object with the reference's config field names (``vision_config``, ``llm_config``, …).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ------------------------------------------------------------------------------------------------
# InternViT
# ------------------------------------------------------------------------------------------------
def rms_norm_cast_then_scale(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """InternRMSNorm / InternLM2RMSNorm (VIT:32-43, LM:129-143): fp32 normalise, cast back to the
    input dtype, THEN multiply by the weight (so the product rounds once more in bf16)."""
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return weight * xf.to(x.dtype)


def vit_pos_embed(pos: Tensor, base_grid: int, gh: int, gw: int) -> Tensor:
    """Position table used by one forward (VIT:87-93,102-105): cls row as-is; patch rows are bicubic
    resized in fp32 (align_corners=False) to the actual grid on EVERY forward, then cast back."""
    dt = pos.dtype
    patch = pos[:, 1:, :].float().reshape(1, base_grid, base_grid, -1).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
    patch = patch.reshape(1, -1, gh * gw).permute(0, 2, 1).to(dt)
    return torch.cat([pos[:, :1, :], patch], dim=1)


def vit_embeddings(sd: SD, cfg, pixel_values: Tensor) -> Tensor:
    """InternVisionEmbeddings.forward (VIT:95-107): conv14/s14 + bias -> tokens (row-major grid) ->
    prepend class token -> add position table.  [F,3,S,S] -> [F, 1+g*g, Hv]."""
    v = cfg.vision_config
    p = "vision_model.embeddings."
    w = sd[p + "patch_embedding.weight"]
    if pixel_values.dim() != 4:
        raise ValueError(f"wrong pixel_values size: {pixel_values.shape}")  # VIT:345
    x = F.conv2d(pixel_values, w, sd[p + "patch_embedding.bias"], stride=v.patch_size)
    nf, _, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "class_embedding"].expand(nf, 1, -1).to(w.dtype)
    x = torch.cat([cls, x], dim=1)
    pos = vit_pos_embed(sd[p + "position_embedding"], v.image_size // v.patch_size, gh, gw)
    return x + pos.to(w.dtype)


def vit_norm(sd: SD, cfg, name: str, x: Tensor) -> Tensor:
    """NORM2FN (VIT:60-63): nn.LayerNorm (ViT-300M) or InternRMSNorm (ViT-6B)."""
    v = cfg.vision_config
    if v.norm_type == "layer_norm":
        return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], v.layer_norm_eps)
    return rms_norm_cast_then_scale(x, sd[name + ".weight"], v.layer_norm_eps)


def vit_attention(sd: SD, cfg, prefix: str, x: Tensor) -> Tensor:
    """InternAttention._naive_attn (VIT:143-160): qkv -> (optional full-width QK RMSNorm) ->
    softmax((q*scale) k^T) v in the input dtype -> proj."""
    v = cfg.vision_config
    nf, n, c = x.shape
    nh = v.num_attention_heads
    d = c // nh
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(nf, n, 3, nh, d).permute(2, 0, 3, 1, 4)
    q, k, vv = qkv[0], qkv[1], qkv[2]
    if v.qk_normalization:  # VIT:148-151 — normalised over the full Hv-wide vector, not per head
        q = rms_norm_cast_then_scale(q.transpose(1, 2).flatten(-2, -1), sd[prefix + "q_norm.weight"],
                                     v.layer_norm_eps).view(nf, n, nh, d).transpose(1, 2)
        k = rms_norm_cast_then_scale(k.transpose(1, 2).flatten(-2, -1), sd[prefix + "k_norm.weight"],
                                     v.layer_norm_eps).view(nf, n, nh, d).transpose(1, 2)
    att = (q * (d ** -0.5)) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    y = (att @ vv).transpose(1, 2).reshape(nf, n, c)
    return F.linear(y, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def vit_mlp(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """InternMLP.forward (VIT:192-196): fc1 -> GELU(erf) -> fc2."""
    h = F.linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"])
    h = F.gelu(h)
    return F.linear(h, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"])


def vit_layer(sd: SD, cfg, i: int, x: Tensor) -> Tensor:
    """InternVisionEncoderLayer.forward (VIT:216-228); DropPath is the identity in eval."""
    p = f"vision_model.encoder.layers.{i}."
    x = x + vit_attention(sd, cfg, p + "attn.", vit_norm(sd, cfg, p + "norm1", x)) * sd[p + "ls1"]
    x = x + vit_mlp(sd, p + "mlp.", vit_norm(sd, cfg, p + "norm2", x)) * sd[p + "ls2"]
    return x


def vit_forward(sd: SD, cfg, pixel_values: Tensor, select_layer: int = -1) -> Tensor:
    """InternVisionModel.forward + InternVisionEncoder.forward (VIT:250-294,324-362).  Returns
    last_hidden_state (select_layer -1) or hidden_states[select_layer] (CHAT:509-518).  There is no
    final norm after the last layer."""
    x = vit_embeddings(sd, cfg, pixel_values)
    states = [x]
    for i in range(cfg.vision_config.num_hidden_layers):
        x = vit_layer(sd, cfg, i, x)
        states.append(x)
    return x if select_layer == -1 else states[select_layer]


# ------------------------------------------------------------------------------------------------
# pixel shuffle + projectors
# ------------------------------------------------------------------------------------------------
def pixel_shuffle_v2(x: Tensor, scale: float = 0.5) -> Tensor:
    """InternVLChatModel.pixel_shuffle, ps_version 'v2' (CHAT:492-506), as its index map: output
    token (i2, j2) = concat over (di, dj) in {(0,0),(0,1),(1,0),(1,1)} of x[2*i2+di, 2*j2+dj, :]."""
    r = int(round(1 / scale))
    n, w, h, c = x.shape
    x = x.reshape(n, w // r, r, h // r, r, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(n, w // r, h // r, r * r * c)


def shuffled_tokens(vit_out: Tensor, scale: float = 0.5) -> Tensor:
    """CHAT:520-527: drop cls, fold to the grid, pixel-shuffle, flatten -> [F, g*g*scale², Hv/scale²]."""
    t = vit_out[:, 1:, :]
    g = int(t.shape[1] ** 0.5)
    t = pixel_shuffle_v2(t.reshape(t.shape[0], g, g, -1), scale)
    return t.reshape(t.shape[0], -1, t.shape[-1])


def projector(sd: SD, name: str, x: Tensor) -> Tensor:
    """mlp1 / motion_mlp (CHAT:238-249): LayerNorm(eps 1e-5) -> Linear -> GELU(erf) -> Linear."""
    x = F.layer_norm(x, (x.shape[-1],), sd[name + ".0.weight"], sd[name + ".0.bias"], 1e-5)
    x = F.linear(x, sd[name + ".1.weight"], sd[name + ".1.bias"])
    x = F.gelu(x)
    return F.linear(x, sd[name + ".3.weight"], sd[name + ".3.bias"])


def extract_feature(sd: SD, cfg, pixel_values: Tensor) -> Tensor:
    """InternVLChatModel.extract_feature (CHAT:508-531): ViT -> shuffled tokens -> mlp1."""
    y = vit_forward(sd, cfg, pixel_values, cfg.select_layer)
    return projector(sd, "mlp1", shuffled_tokens(y, cfg.downsample_ratio))


def slow_pathway_index(t: int) -> Tensor:
    """pack_pathway_output (CHAT:97-133): slow = frames[linspace(0, T-1, T//4).long()], fast = all."""
    return torch.linspace(0, t - 1, t // 4).long()


# ------------------------------------------------------------------------------------------------
# token embedding + scatter
# ------------------------------------------------------------------------------------------------
def scatter_embeds(sd: SD, input_ids: Tensor, img_context_token_id: int, vit_embeds: Tensor,
                   motion_embeds: Optional[Tensor]) -> Tensor:
    """CHAT:324,351-378,398 and CHAT1:268-329 (eval forward, both stages: last <IMG_CONTEXT> of each
    row <- motion token, all others in order <- visual tokens) and CHAT:788-797 (generate(): no motion
    token, every slot <- visual tokens).
    A count mismatch raises (the reference catches it and overwrites a prefix, CHAT:381-386)."""
    emb = F.embedding(input_ids, embed_weight(sd)).clone()
    b, n, c = emb.shape
    emb = emb.reshape(b * n, c)
    sel = input_ids == img_context_token_id
    if motion_embeds is not None:
        csum = torch.cumsum(sel, dim=1)
        last = (csum == csum.max(dim=1, keepdim=True)[0]) & sel
        sel_vis = (sel & ~last).reshape(b * n)
        emb[sel_vis] = emb[sel_vis] * 0.0 + vit_embeds.reshape(-1, c)
        emb[last.reshape(b * n)] = emb[last.reshape(b * n)] * 0.0 + motion_embeds.reshape(-1, c)
    else:
        emb[sel.reshape(b * n)] = vit_embeds.reshape(-1, c)
    return emb.reshape(b, n, c)


# ------------------------------------------------------------------------------------------------
# InternLM2
# ------------------------------------------------------------------------------------------------
def rope_tables(head_dim: int, theta: float, seq_len: int, dtype, max_pos: int = 32768,
                scaling: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """InternLM2RotaryEmbedding / DynamicNTK (LM:161-194,218-243): inv_freq fp32, cos/sin of
    cat(freqs, freqs), CAST TO THE ACTIVATION DTYPE before use (LM:191-194).  The dynamic-NTK base
    change only engages when seq_len > max_position_embeddings (LM:230-235)."""
    base = float(theta)
    if scaling is not None and scaling.get("type") == "dynamic" and seq_len > max_pos:
        f = float(scaling["factor"])
        base = base * ((f * seq_len / max_pos) - (f - 1)) ** (head_dim / (head_dim - 2))
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(seq_len).to(inv_freq.dtype)
    if scaling is not None and scaling.get("type") == "linear":
        t = t / float(scaling["factor"])
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor, position_ids: Tensor) -> Tuple[Tensor, Tensor]:
    """apply_rotary_pos_emb (LM:247-261): x*cos + rotate_half(x)*sin, three rounded ops per tensor."""
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    return (q * c) + (_rot_half(q) * s), (k * c) + (_rot_half(k) * s)


def additive_mask(attention_mask: Tensor, q_len: int, past_len: int, dtype) -> Tensor:
    """_prepare_decoder_attention_mask (LM:844-865, helpers LM:96-125): causal finfo.min mask (only
    when q_len > 1) + expanded padding mask, both additive in the activation dtype."""
    bsz, src = attention_mask.shape
    neg = torch.finfo(dtype).min
    pad = (1.0 - attention_mask[:, None, None, :].expand(bsz, 1, q_len, src).to(dtype))
    pad = pad.masked_fill(pad.to(torch.bool), neg)
    if q_len > 1:
        idx = torch.arange(q_len)
        causal = torch.full((q_len, q_len), neg).masked_fill(idx < (idx + 1).view(q_len, 1), 0).to(dtype)
        if past_len > 0:
            causal = torch.cat([torch.zeros(q_len, past_len, dtype=dtype), causal], dim=-1)
        pad = pad + causal[None, None]
    return pad


# Hook for the build's own fp8 mode (oracle/fp8.py): None = the reference's F.linear.  Called as hook(x, weight, layer, name).
LLM_LINEAR_HOOK = None


# The reference constructor also accepts transformers' LlamaForCausalLM (CHAT:228-229).  transformers is a pip dependency of the reference
# (requirements pin 4.37.2; not vendored in /root/reference), so the Llama branch below restates its published modeling_llama.py: the same
# eager attention, RMSNorm, rotate-half RoPE and SwiGLU MLP as InternLM2 (whose file was derived from it) with q / k / v as three Linears
# and HF's tensor names.  One arithmetic difference exists between transformers versions: 4.37 divides the scores by sqrt(d) as InternLM2
# does, the version installed in the build container (5.x, eager_attention_forward) multiplies them by d ** -0.5 - the fixture
# tests/golden/e2e_llama.pt was recorded from the reference running on the installed version, so LLAMA_SCALE_BY_MULTIPLY defaults to True.
LLAMA_SCALE_BY_MULTIPLY = True


def is_llama(sd: SD) -> bool:
    return "language_model.model.embed_tokens.weight" in sd


def embed_weight(sd: SD) -> Tensor:
    return sd["language_model.model.embed_tokens.weight"] if is_llama(sd) else sd["language_model.model.tok_embeddings.weight"]


def _llm_linear(x: Tensor, w: Tensor, layer: int, name: str) -> Tensor:
    if LLM_LINEAR_HOOK is not None:
        return LLM_LINEAR_HOOK(x, w, layer, name)
    return F.linear(x, w)


def llm_attention(sd: SD, cfg, i: int, x: Tensor, mask: Tensor, position_ids: Tensor,
                  past: Optional[Tuple[Tensor, Tensor]] = None):
    """InternLM2Attention.forward, eager (LM:355-440).  wqkv output features are ordered
    (kv_head, slot, d) with slots 0..g-1 = that group's q heads, slot g = K, slot g+1 = V (LM:375-385)."""
    l = cfg.llm_config
    b, n, _ = x.shape
    nh, nkv, d = l.num_attention_heads, l.num_key_value_heads, l.head_dim
    g = nh // nkv
    llama = is_llama(sd)
    if llama:   # LlamaAttention.forward: q_proj / k_proj / v_proj, heads in order (kv head j serves q heads j g .. j g + g - 1: repeat_kv)
        p = f"language_model.model.layers.{i}.self_attn."
        q = _llm_linear(x, sd[p + "q_proj.weight"], i, "q_proj").view(b, n, nh, d).transpose(1, 2)
        k = _llm_linear(x, sd[p + "k_proj.weight"], i, "k_proj").view(b, n, nkv, d).transpose(1, 2)
        v = _llm_linear(x, sd[p + "v_proj.weight"], i, "v_proj").view(b, n, nkv, d).transpose(1, 2)
    else:
        p = f"language_model.model.layers.{i}.attention."
        qkv = _llm_linear(x, sd[p + "wqkv.weight"], i, "wqkv").view(b, n, nkv, g + 2, d)
        q = qkv[..., :g, :].reshape(b, n, nh, d).transpose(1, 2)
        k = qkv[..., g, :].transpose(1, 2)
        v = qkv[..., g + 1, :].transpose(1, 2)
    kv_len = n + (past[0].shape[-2] if past is not None else 0)
    cos, sin = rope_tables(d, l.rope_theta, kv_len, v.dtype, l.max_position_embeddings, l.rope_scaling)
    q, k = apply_rope(q, k, cos, sin, position_ids)
    if past is not None:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    present = (k, v)
    kr = k[:, :, None].expand(b, nkv, g, kv_len, d).reshape(b, nh, kv_len, d)  # repeat_kv LM:282-291
    vr = v[:, :, None].expand(b, nkv, g, kv_len, d).reshape(b, nh, kv_len, d)
    if llama and LLAMA_SCALE_BY_MULTIPLY:
        w = torch.matmul(q, kr.transpose(2, 3)) * (d ** -0.5)
    else:
        w = torch.matmul(q, kr.transpose(2, 3)) / math.sqrt(d)
    w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    y = torch.matmul(w, vr).transpose(1, 2).contiguous().reshape(b, n, nh * d)
    if llama:
        return _llm_linear(y, sd[p + "o_proj.weight"], i, "o_proj"), present
    return _llm_linear(y, sd[p + "wo.weight"], i, "wo"), present


def llm_mlp(sd: SD, i: int, x: Tensor) -> Tensor:
    """InternLM2MLP.forward (LM:264-278): w2(silu(w1 x) * w3 x)."""
    if is_llama(sd):   # LlamaMLP.forward: down_proj(act(gate_proj(x)) * up_proj(x))
        p = f"language_model.model.layers.{i}.mlp."
        return _llm_linear(F.silu(_llm_linear(x, sd[p + "gate_proj.weight"], i, "gate_proj")) * _llm_linear(x, sd[p + "up_proj.weight"], i, "up_proj"),
                           sd[p + "down_proj.weight"], i, "down_proj")
    p = f"language_model.model.layers.{i}.feed_forward."
    return _llm_linear(F.silu(_llm_linear(x, sd[p + "w1.weight"], i, "w1")) * _llm_linear(x, sd[p + "w3.weight"], i, "w3"),
                       sd[p + "w2.weight"], i, "w2")


def llm_layer(sd: SD, cfg, i: int, x: Tensor, mask: Tensor, position_ids: Tensor, past=None):
    """InternLM2DecoderLayer.forward (LM:635-695)."""
    p = f"language_model.model.layers.{i}."
    eps = cfg.llm_config.rms_norm_eps
    n1, n2 = ("input_layernorm.weight", "post_attention_layernorm.weight") if is_llama(sd) else ("attention_norm.weight", "ffn_norm.weight")
    a, present = llm_attention(sd, cfg, i, rms_norm_cast_then_scale(x, sd[p + n1], eps),
                               mask, position_ids, past)
    x = x + a
    x = x + llm_mlp(sd, i, rms_norm_cast_then_scale(x, sd[p + n2], eps))
    return x, present


def llm_forward(sd: SD, cfg, inputs_embeds: Tensor, attention_mask: Optional[Tensor] = None,
                position_ids: Optional[Tensor] = None, past: Optional[List] = None,
                all_hidden: bool = False):
    """InternLM2Model.forward (LM:868-998): default position_ids = arange(past, past+n); layers; final
    RMSNorm.  Returns (normed hidden, presents, [layer inputs..., normed hidden] if all_hidden)."""
    b, n, _ = inputs_embeds.shape
    past_len = past[0][0].shape[2] if past is not None else 0
    if position_ids is None:
        position_ids = torch.arange(past_len, n + past_len, dtype=torch.long).unsqueeze(0)
    if attention_mask is None:
        attention_mask = torch.ones((b, n + past_len), dtype=torch.bool)
    mask = additive_mask(attention_mask, n, past_len, inputs_embeds.dtype)
    x = inputs_embeds
    states, presents = [], []
    for i in range(cfg.llm_config.num_hidden_layers):
        if all_hidden:
            states.append(x)
        x, pr = llm_layer(sd, cfg, i, x, mask, position_ids, past[i] if past is not None else None)
        presents.append(pr)
    x = rms_norm_cast_then_scale(x, sd["language_model.model.norm.weight"], cfg.llm_config.rms_norm_eps)
    if all_hidden:
        states.append(x)
    return x, presents, states


def lm_logits(sd: SD, hidden: Tensor) -> Tensor:
    """InternLM2ForCausalLM.forward lm-head (LM:1094-1096): matmul in the model dtype, THEN .float()."""
    return F.linear(hidden, sd["language_model.lm_head.weight"] if is_llama(sd) else sd["language_model.output.weight"]).float()


def score_head(sd: SD, cfg, x: Tensor) -> Tensor:
    """MLP.forward (CHAT:43-94): Linear+ReLU chain 4096->1024->256->64->16->1, ReLU after EVERY layer
    including the last; ``ln1`` exists in the state-dict but is unused (CHAT:55,85)."""
    n = 1
    while f"mlpscore.fc{n}.weight" in sd:
        x = F.relu(F.linear(x, sd[f"mlpscore.fc{n}.weight"], sd[f"mlpscore.fc{n}.bias"]))
        n += 1
    return x


# ------------------------------------------------------------------------------------------------
# end-to-end passes
# ------------------------------------------------------------------------------------------------
def forward_eval(sd: SD, cfg, pixel_values: Tensor, input_ids: Tensor, attention_mask: Optional[Tensor],
                 image_flags: Tensor, labels: Tensor, motion_feature: Optional[Tensor],
                 img_context_token_id: int, mos: Optional[Tensor] = None, stage: int = 2,
                 return_intermediates: bool = False) -> Dict[str, Tensor]:
    """InternVLChatModel.forward, stage-2 flavour (CHAT:306-488) or stage-1 flavour (CHAT1:331-366).

    stage 2 -> {'loss' (L1 vs mos), 'score1' [B], 'label' [B*(N-1)], 'logit' [B*(N-1)]}
    stage 1 -> {'label', 'logit'} — same pass including the motion token (CHAT1:278-329), no score head.
    ``motion_feature`` [B, motion_dim] replaces the SlowFast branch (CHAT:337-344) — an input on both
    sides."""
    flags = image_flags.squeeze(-1)
    vit_embeds = extract_feature(sd, cfg, pixel_values)
    vit_embeds = vit_embeds[flags == 1]
    motion_embeds = projector(sd, "motion_mlp", motion_feature.view(input_ids.shape[0], -1))
    emb = scatter_embeds(sd, input_ids, img_context_token_id, vit_embeds, motion_embeds)
    hidden, _, _ = llm_forward(sd, cfg, emb, attention_mask)
    logits = lm_logits(sd, hidden)
    shift_logits = logits[..., :-1, :].contiguous().view(-1, logits.shape[-1])
    shift_labels = labels[..., 1:].contiguous().view(-1)
    out = {"label": shift_labels, "logit": torch.argmax(shift_logits, dim=1)}
    if stage == 2:
        x = hidden[:, -4, :]  # CHAT:469-470 (hidden_states[-1] is the post-final-norm state, LM:984-988)
        if torch.isnan(x).any():  # CHAT:471-473
            x = torch.nan_to_num(x, nan=0.0, posinf=1e9, neginf=-1e9)
        score = score_head(sd, cfg, x).squeeze(1)
        out["score1"] = score
        if mos is not None:
            out["loss"] = F.l1_loss(score, mos)
    if return_intermediates:
        out.update(vit_embeds=vit_embeds, motion_embeds=motion_embeds, inputs_embeds=emb, hidden=hidden,
                   logits=logits)
    return out


def answer_slice(labels_row: Tensor, logit_row: Tensor, im_end_id: int) -> Tensor:
    """stage2_eval.py:940-941: answer ids = labels not in {-100, <|im_end|>}; the predicted tokens are
    logit[-len(ans)-1 : -1] of the flattened shifted argmax."""
    ans = labels_row[(labels_row != -100) & (labels_row != im_end_id)]
    return logit_row[-len(ans) - 1:-1]


def greedy_generate(sd: SD, cfg, inputs_embeds: Tensor, attention_mask: Tensor, max_new_tokens: int,
                    eos_token_id: Optional[int] = None) -> Tensor:
    """Greedy decode = the loop HF ``generate`` runs for InternVLChatModel.generate (CHAT:769-811)
    through prepare_inputs_for_generation (LM:1126-1163): embeds on step 0, then the last token id;
    position_ids = cumsum(mask)-1; tuple KV cache concat (LM:397-402).  Batch 1 or equal lengths."""
    b = inputs_embeds.shape[0]
    mask = attention_mask.clone().long()
    pos = (mask.cumsum(-1) - 1).masked_fill(mask == 0, 1)
    hidden, past, _ = llm_forward(sd, cfg, inputs_embeds, mask.bool(), pos)
    out: List[Tensor] = []
    done = torch.zeros(b, dtype=torch.bool)
    for _ in range(max_new_tokens):
        nxt = lm_logits(sd, hidden[:, -1:, :])[:, -1, :].argmax(-1)
        if eos_token_id is not None:
            nxt = torch.where(done, torch.full_like(nxt, eos_token_id), nxt)
        out.append(nxt)
        if eos_token_id is not None:
            done = done | (nxt == eos_token_id)
            if bool(done.all()):
                break
        mask = torch.cat([mask, torch.ones((b, 1), dtype=mask.dtype)], dim=1)
        pos = (mask.cumsum(-1) - 1)[:, -1:]
        emb = F.embedding(nxt[:, None], embed_weight(sd))
        hidden, past, _ = llm_forward(sd, cfg, emb, mask.bool(), pos, past)
    return torch.stack(out, dim=1)


IMAGENET_MEAN = (0.485, 0.456, 0.406)   # internvl/train/constants.py:10-11
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize_frames_u8(frames_hwc: Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD, dtype=torch.bfloat16) -> Tensor:
    """Eval transform after the resize (internvl/train/dataset.py:267-274): ToTensor = uint8 HWC -> float CHW / 255;
    Normalize = (x - mean) / std in fp32 (torchvision semantics: sub_ then div_); cast to bf16 at the call site
    (stage2_eval.py:932).  torchvision is absent here, so this restates its published definition (parity unpinned
    by a run of the reference; the arithmetic is three fp32 ops)."""
    x = frames_hwc.permute(0, 3, 1, 2).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return x.sub_(m).div_(s).to(dtype)


LEVEL_WORDS = ("bad", "poor", "fair", "good", "excellent")


def parse_level(text: str) -> int:
    """stage2_eval.py:956-967: substring test in the reference's order bad, poor, fair, good,
    excellent -> 1..5, else 0."""
    for word, level in (("bad", 1), ("poor", 2), ("fair", 3), ("good", 4), ("excellent", 5)):
        if word in text:
            return level
    return 0
