"""Host-side mirror of the reference's ``InternVLChatModel`` over the gfx950 C-ABI library.

Same class name, constructor, attributes and method signatures as
internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:195-853 (stage-2 flavour; ``stage=1`` gives
internvl_chat_eval1's return dict), so ``stage{1,2}_eval.py`` call sites run unchanged:

    model = InternVLChatModel.from_pretrained(path, torch_dtype=torch.bfloat16, config=cfg)
    model.img_context_token_id = ...; model.eval(); model.cuda()
    out = model(mos=..., pixel_values=..., input_ids=..., attention_mask=..., image_flags=..., labels=...)
    out['score1'], out['logit'], out['label']

PyTorch is used for tensor containers, weight (de)serialisation and index bookkeeping only; every FLOP of
the hot path runs in ``libaigv_amd.so``.  There is no CPU / eager fallback: without the library or a GPU the
hot-path methods raise.

Documented deviations from the reference (all outside what its eval scripts exercise):
  * ``logit`` holds argmax ids only at positions whose shifted label is not -100 (the answer rows the eval
    slices, stage2_eval.py:940-941); other positions are -1 unless ``full_logits=True``.
  * the score row ``hidden[:, -4]`` is taken relative to each clip's true (un-padded) end.
  * a visual-token count mismatch raises instead of overwriting a prefix (modeling_internvl_chat.py:381-386).
  * the SlowFast motion branch (``slowfast_model``) is the native ``SlowFastR50`` when the state dict carries
    ``slowfast_model.*`` tensors (the reference downloads them from pytorchvideo's hub at construction time,
    which cannot happen offline); otherwise pass ``motion_feature=[B, 2304]`` or set ``slowfast_model`` to a
    callable with the reference's interface.
  * ``generate`` implements greedy decoding and multinomial sampling (temperature / top-k / top-p, HF's warper order) with HF's
    repetition_penalty / no_repeat_ngram_size processors; the reference defers to HF ``generate`` - its eval configs use
    do_sample=False.  num_beams > 1 runs HF's beam search (beam.py: best hypothesis only; beam sampling / beam groups raise).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native, synth
from .config import InternVLChatConfig
from .conversation import get_conv_template


# ------------------------------------------------------------------------------------------------------
# host-side weight preparation (not on the hot path: runs once per weight upload)
# ------------------------------------------------------------------------------------------------------
def rope_tables(head_dim: int, theta: float, n_pos: int, max_pos: int = 32768, scaling: Optional[dict] = None,
                seq_len: Optional[int] = None):
    """cos/sin [n_pos, head_dim/2] bf16, computed as the reference does (modeling_internlm2.py:161-243):
    fp32 inv_freq and angles, cos/sin in fp32, then cast to the activation dtype.  The reference table is
    cat(freqs, freqs) so only the first half is stored.

    ``n_pos`` is only the number of table rows (a capacity: packed tokens of a batch, KV capacity).  Dynamic-NTK
    rescaling is decided by ``seq_len``, the length of the longest SINGLE sequence of the call - the reference keys it
    on the per-sequence ``kv_seq_len`` (:218-243, :387-391), so a batch of many short clips is never rescaled however
    many tokens it packs.  ``seq_len=None`` means "no sequence is longer than max_pos" (no rescale)."""
    base = float(theta)
    if scaling is not None and scaling.get("type") == "dynamic" and seq_len is not None and seq_len > max_pos:
        f = float(scaling["factor"])
        base = base * ((f * seq_len / max_pos) - (f - 1)) ** (head_dim / (head_dim - 2))
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(n_pos).to(inv_freq.dtype)
    if scaling is not None and scaling.get("type") == "linear":
        t = t / float(scaling["factor"])
    freqs = torch.outer(t, inv_freq)
    return freqs.cos().to(torch.bfloat16).contiguous(), freqs.sin().to(torch.bfloat16).contiguous()


def resized_pos_table(pos: torch.Tensor, base_grid: int, grid: int) -> torch.Tensor:
    """Position table for a ``grid x grid`` patch grid (modeling_intern_vit.py:87-93,102-105): class row
    as-is, patch rows bicubic-resized in fp32 (align_corners=False) and cast back.  Precomputed once per
    upload instead of on every forward; at the native grid it is the identity."""
    pos = pos.detach().to("cpu")
    dt = pos.dtype
    patch = pos[:, 1:, :].float().reshape(1, base_grid, base_grid, -1).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, size=(grid, grid), mode="bicubic", align_corners=False)
    patch = patch.reshape(1, -1, grid * grid).permute(0, 2, 1).to(dt)
    return torch.cat([pos[:, :1, :], patch], dim=1).contiguous()


# ------------------------------------------------------------------------------------------------------
# parameter containers with the reference's module paths (so state_dict keys match §8a row W)
# ------------------------------------------------------------------------------------------------------
class _Node(nn.Module):
    """Parameter holder; children are added under their reference names."""

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def __getitem__(self, i):
        return self._modules[str(i)]


class _VisionModel(_Node):
    """``model.vision_model`` surface used by the drivers (modeling_intern_vit.py:297-323)."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def resize_pos_embeddings(self, old_size, new_size, patch_size):
        # modeling_intern_vit.py:309-319 (weight surgery, host side)
        emb = self.embeddings
        pos = emb.position_embedding.data
        new = resized_pos_table(pos, old_size // patch_size, new_size // patch_size).to(pos.device)
        emb.position_embedding = nn.Parameter(new, requires_grad=False)
        self._owner.config.vision_config.image_size = new_size
        self._owner._invalidate()

    def get_input_embeddings(self):
        return self.embeddings


class _LanguageModel(_Node):
    """``model.language_model`` surface (modeling_internlm2.py:1016-1032 + HF resize_token_embeddings)."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    @property
    def config(self):
        return self._owner.config.llm_config

    def get_input_embeddings(self):
        return self.model.tok_embeddings

    def get_output_embeddings(self):
        return self.output

    def resize_token_embeddings(self, n: int):
        for node in (self.model.tok_embeddings, self.output):
            old = node.weight.data
            new = torch.zeros((n, old.shape[1]), dtype=old.dtype, device=old.device)
            new[: min(n, old.shape[0])] = old[: min(n, old.shape[0])]
            if n > old.shape[0]:
                new[old.shape[0]:].normal_(0.0, self.config.initializer_range)
            node.weight = nn.Parameter(new, requires_grad=False)
        self.config.vocab_size = n
        self._owner._invalidate()
        return self.model.tok_embeddings


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


_PARKED_GRAPHS: list = []      # captured passes that were dropped: kept alive until the interpreter exits (InternVLChatModel._drop_graphs)


# ------------------------------------------------------------------------------------------------------
class VisualAhead:
    """The visual front of a LATER ``forward`` call, started ahead of time by ``InternVLChatModel.prefetch``: pre-projector tokens
    [F, ntok, 4 Hv], the SlowFast feature of the clips (or None) and the event the consuming stream waits for.  Pass it as ``pixel_values``."""
    __slots__ = ("tokens", "motion", "event", "n_clips")

    def __init__(self, tokens, motion, event, n_clips):
        self.tokens, self.motion, self.event, self.n_clips = tokens, motion, event, n_clips


class InternVLChatModel(nn.Module):
    main_input_name = "pixel_values"

    def __init__(self, config: InternVLChatConfig, vision_model=None, language_model=None, use_flash_attn=True,
                 device=None, dtype=torch.bfloat16, stage: int = 2, max_clips: int = 4, max_frames: Optional[int] = None,
                 max_tokens: int = 0):
        super().__init__()
        if vision_model is not None or language_model is not None:
            raise NotImplementedError("pass weights through load_state_dict / from_pretrained")
        if config.llm_config.architectures[0] not in ("InternLM2ForCausalLM", "LlamaForCausalLM"):
            # the two families the reference constructor accepts (modeling_internvl_chat.py:228-233).  A Llama checkpoint is re-packed
            # into the InternLM2 weight layout when it is loaded (weights.llama_to_internlm2): the kernels and this module's parameter
            # names are the same for both families
            raise NotImplementedError(f"{config.llm_config.architectures[0]} is not implemented.")
        if dtype != torch.bfloat16:
            raise NotImplementedError("the gfx950 path computes in bf16 (the reference eval dtype, stage2_eval.py:780)")
        self.config = config
        self.stage = stage
        self.patch_size = config.vision_config.patch_size
        self.select_layer = config.select_layer
        self.template = config.template
        self.num_image_token = config.num_image_token
        self.downsample_ratio = config.downsample_ratio
        self.ps_version = config.ps_version
        if self.ps_version != "v2":
            raise NotImplementedError("only ps_version 'v2' (the shipped config) is on the hot path")
        self.llm_arch_name = config.llm_config.architectures[0]
        self.img_context_token_id = None
        self.conv_template = get_conv_template(self.template)
        self.system_message = self.conv_template.system_message
        self.slowfast_model = None          # optional callable([slow, fast]) -> [B, 2304, 1, 1, 1]
        self._max_clips, self._max_frames, self._max_tokens = max_clips, max_frames, max_tokens
        self._ctx = None
        self._ctx_key = None
        self._dirty = True
        self._rope_ntk = 0                  # sequence length the dynamic-NTK rotary base is currently built for (0: plain tables)

        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.vision_model = _VisionModel(self)
        self.language_model = _LanguageModel(self)
        for name, shape, _kind in synth.weight_shapes(config):
            if stage == 1 and name.startswith("mlpscore."):
                continue
            root, rest = name.split(".", 1)
            if root not in self._modules:
                self.add_module(root, _Node())
            _attach(self._modules[root], rest, torch.empty(shape, dtype=dtype, device=dev))

    # ---- construction / (de)serialisation ----------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, torch_dtype=torch.bfloat16, config: Optional[InternVLChatConfig] = None, **kw):
        """Load ``config.json`` + the MODEL shards of a checkpoint directory with the reference's state-dict names
        (stage2_eval.py:779-780); ``slowfast_model.*`` tensors build the native motion branch.

        A directory written by the reference trainer (HF Trainer) also holds ``training_args.bin``, ``optimizer.pt``,
        ``scheduler.pt``, ``rng_state*.pth`` and possibly ``lora_weights.pth`` (stage2_train.py:223-235): only
        ``model*.safetensors`` / ``pytorch_model*.bin`` are read (through the ``*.index.json`` weight map when there is one),
        ``lora_weights.pth`` is folded in by ``weights.merge_lora_state_dict``, everything else is ignored."""
        if config is None:
            config = InternVLChatConfig.from_pretrained(path)
        model = cls(config, dtype=torch_dtype, **kw)
        model.load_state_dict(cls._read_checkpoint(path))
        return model

    @staticmethod
    def _checkpoint_files(path) -> List[str]:
        """The weight shards of a checkpoint directory, in load order (host logic; no tensor is read)."""
        import json
        names = sorted(os.listdir(path))
        for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
            if index in names:
                with open(os.path.join(path, index)) as f:
                    shards = sorted(set(json.load(f)["weight_map"].values()))
                missing = [x for x in shards if x not in names]
                if missing:
                    raise FileNotFoundError(f"{index} names shards that are not under {path}: {missing}")
                return shards
        st = [f for f in names if f.endswith(".safetensors") and (f.startswith("model") or f.startswith("pytorch_model"))]
        if st:
            return st
        return [f for f in names if f.startswith("pytorch_model") and f.endswith(".bin")]

    @classmethod
    def _read_checkpoint(cls, path) -> Dict[str, torch.Tensor]:
        files = cls._checkpoint_files(path)
        if not files:
            raise FileNotFoundError(f"no model shards (model*.safetensors / pytorch_model*.bin) found under {path}")
        sd: Dict[str, torch.Tensor] = {}
        for f in files:
            fp = os.path.join(path, f)
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file
                sd.update(load_file(fp))
            else:
                sd.update(torch.load(fp, map_location="cpu", weights_only=True))
        lora = os.path.join(path, "lora_weights.pth")
        has_adapters = any(".lora_A." in k for k in sd)
        if os.path.exists(lora) or has_adapters:
            from .weights import merge_lora_state_dict
            extra = torch.load(lora, map_location="cpu", weights_only=True) if os.path.exists(lora) else None
            sd = merge_lora_state_dict(sd, extra)
        return sd

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sf = {k: v for k, v in state_dict.items() if k.startswith("slowfast_model.")}
        if sf:   # the motion branch's backbone travels in the reference's checkpoints (modeling_internvl_chat.py:253)
            from .slowfast import SlowFastR50
            self.slowfast_model = SlowFastR50(sf)
        sd = {k: v for k, v in state_dict.items() if not k.startswith("slowfast_model.")}
        sd = self._family_names(sd)
        if self.stage == 1:
            sd = {k: v for k, v in sd.items() if not k.startswith("mlpscore.")}
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                               f"unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        with torch.no_grad():
            for k, p in own.items():
                if k in sd:
                    if tuple(sd[k].shape) != tuple(p.shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(p.shape)}")
                    p.copy_(sd[k].to(p.dtype))
        self._invalidate()
        return missing, unexpected

    def _family_names(self, sd):
        """A transformers-Llama state dict (the reference's second LLM family) -> this module's InternLM2-layout names and packing."""
        from . import weights
        if not weights.is_llama_state_dict(sd):
            return sd
        if self.llm_arch_name != "LlamaForCausalLM":
            raise RuntimeError("load_state_dict: Llama tensor names in a checkpoint for an InternLM2 configuration")
        return weights.llama_to_internlm2(sd, self.config.llm_config)

    def load_state_dict_stream(self, named_tensors, strict: bool = True):
        """load_state_dict from an iterable of (name, tensor) without ever holding the whole state dict on the host (InternVL2-26B: 51 GB):
        every tensor is copied into its parameter as it arrives.  The same contract as load_state_dict: InternLM2-layout names always load,
        transformers-Llama names are re-packed on the fly (Llama configurations only), ``slowfast_model.*`` tensors build the motion branch,
        names the model does not own raise, and so do missing tensors unless ``strict=False`` (then their names are returned).  Whatever
        happens, the native copy of the weights is invalidated - a failed stream never leaves it silently out of step with the module."""
        from .weights import llama_stream_to_internlm2
        own = dict(self.named_parameters())
        seen = set()
        slowfast = {}

        def routed():
            for k, v in named_tensors:
                if k.startswith("slowfast_model."):
                    slowfast[k] = v
                    continue
                yield k, v

        stream = routed()
        if self.llm_arch_name == "LlamaForCausalLM":
            stream = llama_stream_to_internlm2(stream, self.config.llm_config)     # (InternLM2-layout names pass through unchanged)
        try:
            with torch.no_grad():
                for k, v in stream:
                    if self.stage == 1 and k.startswith("mlpscore."):
                        continue
                    if k not in own:
                        raise RuntimeError(f"load_state_dict_stream: unexpected tensor {k}")
                    if tuple(v.shape) != tuple(own[k].shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(own[k].shape)}")
                    own[k].copy_(v.to(own[k].dtype))
                    seen.add(k)
            if slowfast:
                from .slowfast import SlowFastR50
                self.slowfast_model = SlowFastR50(slowfast)
        finally:
            self._invalidate()
        missing = [k for k in own if k not in seen]
        if strict and missing:
            raise RuntimeError(f"load_state_dict_stream: missing {missing[:5]}{'...' if len(missing) > 5 else ''}")
        return missing

    def _apply(self, fn, *a, **k):  # .cuda() / .to(): weights move, the native copy must follow
        out = super()._apply(fn, *a, **k)
        self._invalidate()
        return out

    def _invalidate(self):
        self._dirty = True
        self._drop_graphs()

    @property
    def device(self):
        return self.mlp1._modules["1"].weight.device

    @property
    def dtype(self):
        return torch.bfloat16

    # ---- native context ---------------------------------------------------------------------------------
    def _rope_seq_len(self, seq_len: int) -> int:
        """The sequence length the dynamic-NTK base is computed for, 0 = plain tables.  The reference rescales the rotary
        base when the per-sequence ``kv_seq_len`` (the padded N of the call, plus any cache) exceeds
        ``max_position_embeddings`` (modeling_internlm2.py:218-243); the packed token count of a batch plays no part."""
        l = self.config.llm_config
        sc = l.rope_scaling
        if sc and sc.get("type") == "dynamic" and seq_len > l.max_position_embeddings:
            return int(seq_len)
        return 0

    def _native(self, n_frames: int = 0, n_tokens: int = 0, n_clips: int = 0, out_rows: int = 0, kv_cap: int = 0,
                seq_len: int = 0):
        """Create (or grow) the native context and upload weights if they changed.  ``seq_len`` = the longest single
        sequence of the call about to run (0: leave the rotary tables as they are)."""
        if self.device.type != "cuda":
            raise native.NativeError("the scorer hot path runs on an MI355X only: move the model with .cuda() "
                                     "(there is no CPU fallback)")
        lib = native.load()
        cfg, v, l = self.config, self.config.vision_config, self.config.llm_config
        key = getattr(self, "_cap", None) or dict(frames=0, tokens=0, clips=0, rows=0, kv=0)
        # capacities only grow, in coarse steps (tokens by 512, KV by 256, output rows by 64).  A request above one re-allocates the
        # workspaces of the context (aigv_ctx_resize: a device sync and a few hipMallocs); the weights are uploaded once per load
        up = lambda x, m: (int(x) + m - 1) // m * m
        want = dict(frames=max(key["frames"], n_frames, self._max_frames or 0, 1),
                    tokens=max(key["tokens"], up(max(n_tokens, self._max_tokens, 1), 512)), clips=max(key["clips"], n_clips, self._max_clips, 1),
                    rows=max(key["rows"], up(max(out_rows, 64), 64)), kv=max(key["kv"], up(kv_cap, 256)))
        geom = (v.image_size if cfg.force_image_size is None else cfg.force_image_size, v.hidden_size, l.vocab_size,
                self.select_layer)
        if self._ctx is None or want != key or geom != self._ctx_key:
            self._drop_graphs()                  # captured graphs hold the old workspaces' addresses
            # same model, larger capacities: only the workspaces are re-allocated (aigv_ctx_resize), the weights stay on the device
            grow = self._ctx is not None and geom == self._ctx_key and not self._dirty
            if self._ctx is not None and not grow:
                lib.aigv_ctx_destroy(self._ctx)
                self._ctx = None
            c = native.AigvConfig()
            c.vit_hidden, c.vit_inter, c.vit_heads, c.vit_layers = v.hidden_size, v.intermediate_size, v.num_attention_heads, v.num_hidden_layers
            c.image_size, c.patch_size, c.num_channels = cfg.image_size, v.patch_size, v.num_channels
            c.vit_norm_rms = 1 if v.norm_type == "rms_norm" else 0
            c.vit_qk_norm, c.vit_qkv_bias, c.vit_eps = int(v.qk_normalization), int(v.qkv_bias), v.layer_norm_eps
            c.select_layer, c.shuffle = self.select_layer, int(round(1 / cfg.downsample_ratio))
            c.llm_hidden, c.llm_inter, c.llm_heads, c.llm_kv_heads = l.hidden_size, l.intermediate_size, l.num_attention_heads, l.num_key_value_heads
            c.llm_layers, c.vocab, c.rms_eps = l.num_hidden_layers, l.vocab_size, l.rms_norm_eps
            c.max_positions = max(want["tokens"], want["kv"], 64)
            c.motion_dim = cfg.motion_dim
            dims = list(cfg.score_dims) if self.stage == 2 else [1]
            c.n_score_layers = len(dims)
            for i, d in enumerate(dims):
                c.score_dims[i] = d
            c.max_frames = want["frames"]
            c.vit_chunk = min(want["frames"], 64)
            c.max_tokens, c.max_seqs, c.max_out_rows, c.kv_capacity = want["tokens"], want["clips"], want["rows"], want["kv"]
            if grow:
                rc = lib.aigv_ctx_resize(self._ctx, C.byref(c))
                if rc != 0:                      # e.g. out of memory: the context is unusable now
                    msg = lib.aigv_last_error(self._ctx)
                    lib.aigv_ctx_destroy(self._ctx)
                    self._ctx, self._dirty = None, True
                    raise native.NativeError(f"libaigv_amd error {rc}: {msg.decode() if msg else '?'}")
                self._cap = want
                if c.max_positions != self._n_pos:   # longer rotary tables: reload them (aigv_finalize_weights inside; it keeps the precision mode)
                    self._n_pos = c.max_positions
                    self._upload_rope()
            else:
                h = C.c_void_p()
                native.check(lib.aigv_ctx_create(self.device.index or 0, C.byref(c), C.byref(h)))
                self._ctx, self._cap, self._ctx_key, self._dirty = h, want, geom, True
                self._n_pos = c.max_positions
        if seq_len:
            ntk = self._rope_seq_len(seq_len)
            if ntk != getattr(self, "_rope_ntk", 0):
                self._rope_ntk = ntk
                self._drop_graphs()              # (the rotary tables a captured pass reads are replaced)
                if not self._dirty:
                    self._upload_rope()
        if self._dirty:
            self._upload()
        return lib, self._ctx

    def _rope_for_decode(self, kv_seq_len: int):
        """Dynamic-NTK rope scaling during decode: the reference's rotary module rebuilds its tables, with the base of the CURRENT
        ``kv_seq_len`` (cached keys + the new token, the padded width of the batch), whenever that exceeds what it has cached - i.e.
        at every decode step past ``max_position_embeddings`` - and rotates only the new token's q / k with them; cached keys keep the
        base they were rotated with (modeling_internlm2.py:187-194,227-243).  Here: the tables are rebuilt and swapped before such a step
        (a host computation and two H2D copies per token - this far out, decode is not a throughput path)."""
        ntk = self._rope_seq_len(kv_seq_len)
        if ntk != getattr(self, "_rope_ntk", 0):
            self._rope_ntk = ntk
            self._drop_graphs()
            torch.cuda.current_stream(self.device).synchronize()     # the previous step still reads the tables being replaced
            self._upload_rope()

    def _upload_rope(self):
        """(Re)build the rotary tables: rows = the context's position capacity, base = the dynamic-NTK base of the current call."""
        lib, ctx = native.load(), self._ctx
        l = self.config.llm_config
        ntk = getattr(self, "_rope_ntk", 0)
        cos, sin = rope_tables(l.head_dim, l.rope_theta, self._n_pos, l.max_position_embeddings, l.rope_scaling, seq_len=ntk or None)
        for name, t in (("rope.cos", cos), ("rope.sin", sin)):
            shape = (C.c_int64 * t.dim())(*t.shape)
            native.check(lib.aigv_load_weight(ctx, name.encode(), t.data_ptr(), shape, t.dim(), 0, 0), ctx)
        if not self._dirty:     # tables swapped under finalized weights: re-derive the pointers.  The library keeps the precision mode
            native.check(lib.aigv_finalize_weights(ctx), ctx)     # and the e4m3 copies: no InternLM2 linear was reloaded (aigv_amd.h)

    def _upload(self):
        lib, ctx = native.load(), self._ctx
        cfg, v, l = self.config, self.config.vision_config, self.config.llm_config

        def put(name, t):
            t = t.detach()
            if t.dtype != torch.bfloat16:
                t = t.to(torch.bfloat16)
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            native.check(lib.aigv_load_weight(ctx, name.encode(), t.data_ptr(), shape, t.dim(), 0, int(t.is_cuda)), ctx)

        for name, p in self.named_parameters():
            if name.startswith("mlpscore.ln1"):
                continue  # present in the reference state-dict, unused by its forward (:55,85)
            if name == "vision_model.embeddings.position_embedding":
                put(name, resized_pos_table(p.data, v.image_size // v.patch_size, cfg.image_size // v.patch_size))
            else:
                put(name, p.data)
        if self.stage == 1:  # stage-1 flavour has no score head: a 1-wide dummy keeps the ABI uniform
            put("mlpscore.fc1.weight", torch.zeros(1, l.hidden_size, dtype=torch.bfloat16))
            put("mlpscore.fc1.bias", torch.zeros(1, dtype=torch.bfloat16))
        self._upload_rope()
        native.check(lib.aigv_finalize_weights(ctx), ctx)
        # finalize resets the context to bf16 and drops stale e4m3 weight copies: a re-created context or reloaded weights keep the mode
        native.check(lib.aigv_set_precision(ctx, 1 if getattr(self, "_precision", "bf16") == "fp8" else 0), ctx)
        # per-context switches survive a re-created context
        native.check(lib.aigv_set_gemm_mode(ctx, int(getattr(self, "_gemm_mode", -1))), ctx)
        native.check(lib.aigv_set_row_trimming(ctx, int(getattr(self, "_row_trim", True))), ctx)
        native.check(lib.aigv_set_attention_numerics(ctx, int(getattr(self, "_attn_numerics", 0))), ctx)
        self._dirty = False

    def __del__(self):
        try:
            _PARKED_GRAPHS.extend(v for v in getattr(self, "_graphs", {}).values() if isinstance(v, tuple))      # (see _drop_graphs)
        except Exception:
            pass
        try:
            if getattr(self, "_ctx", None) is not None:
                native.release("aigv_ctx_destroy", self._ctx)      # (parked while a stream capture is underway: native.release)
        except Exception:
            pass

    # ---- hot path -----------------------------------------------------------------------------------------
    def ingest_frames(self, frames_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                      size: Optional[int] = None) -> torch.Tensor:
        """uint8 [F, H, W, 3] RGB frames (one tensor, or a list of per-clip tensors) -> normalised bf16 NCHW ``pixel_values`` on the GPU: the reference's per-frame
        ``image.resize((448, 448))`` (PIL BICUBIC; dataset.py:702-738 with max_num = 1, stage2_eval.py:453-456) when the
        frames are not at the model resolution yet, then ToTensor + Normalize + the bf16 cast of its eval transform
        (dataset.py:267-274, stage2_eval.py:932).  The resize is bit-exact with Pillow (include/aigv_amd.h)."""
        parts = list(frames_u8) if isinstance(frames_u8, (list, tuple)) else [frames_u8]
        for t in parts:
            if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[-1] != 3 or t.shape[1:] != parts[0].shape[1:]:
                raise ValueError("frames must be uint8 [F, H, W, 3] (or a list of such tensors of one frame size)")
        lib = native.load()
        if self.device.type != "cuda":
            raise native.NativeError("the scorer hot path runs on an MI355X only (no CPU fallback)")
        S = int(size or self.config.image_size)
        # (pinned host frames go up without blocking the host: the copy is ordered on the current stream like the kernels that read it.  A list -
        # the clips of one group, each in its own host buffer - is copied clip by clip and joined on the device: no host-side concatenation)
        parts = [t.to(self.device, non_blocking=not t.is_cuda and t.is_pinned()) for t in parts]
        f = (torch.cat(parts) if len(parts) > 1 else parts[0]).contiguous()
        n, h, w, _ = f.shape
        out = torch.empty((n, 3, S, S), dtype=torch.bfloat16, device=self.device)
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        if (h, w) == (S, S):
            native.check(lib.aigv_op_frame_ingest(f.data_ptr(), n, h, w, m3, s3, out.data_ptr(), native.stream_ptr()))
        else:
            tmp = torch.empty(n * h * S * 3, dtype=torch.uint8, device=self.device)
            native.check(lib.aigv_op_frame_resize_ingest(f.data_ptr(), n, h, w, S, S, m3, s3, tmp.data_ptr(), None, out.data_ptr(),
                                                         native.stream_ptr()))
        return out

    def vit_tokens(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """InternViT -> drop cls -> pixel-shuffle: [F,3,S,S] -> [F, ntok, 4*Hv] pre-projector tokens (the
        frame-DP all-gather payload; modeling_internvl_chat.py:509-527)."""
        if pixel_values.dim() != 4:
            raise ValueError(f"wrong pixel_values size: {pixel_values.shape}")  # modeling_intern_vit.py:345
        nf = pixel_values.shape[0]
        S = self.config.image_size
        if tuple(pixel_values.shape[1:]) != (self.config.vision_config.num_channels, S, S):
            raise ValueError(f"pixel_values must be [F,{self.config.vision_config.num_channels},{S},{S}], got {tuple(pixel_values.shape)}")
        lib, ctx = self._native(n_frames=nf)
        self._wait_for_prefetch()
        pv = pixel_values.to(device=self.device, dtype=torch.bfloat16).contiguous()
        out = torch.empty((nf, self.num_image_token, self.config.proj_in), dtype=torch.bfloat16, device=self.device)
        native.check(lib.aigv_vit_forward(ctx, pv.data_ptr(), nf, out.data_ptr(), native.stream_ptr()), ctx)
        return out

    def _take_ahead(self, pixel_values, visual_tokens, motion_feature):
        """``pixel_values`` may be the handle of a visual front started ahead of time (``prefetch``): wait for it on the caller's stream and continue
        from its tokens / SlowFast feature; a plain ``pixel_values`` first waits for any prefetch in flight (one visual front at a time)."""
        if isinstance(pixel_values, VisualAhead):
            torch.cuda.current_stream(self.device).wait_event(pixel_values.event)
            return None, pixel_values.tokens, pixel_values.motion if motion_feature is None else motion_feature
        if pixel_values is not None:
            self._wait_for_prefetch()
        return pixel_values, visual_tokens, motion_feature

    def _wait_for_prefetch(self):
        """The InternViT workspaces of the context serve ONE visual front at a time: a pass that runs the ViT on the caller's stream (eager or
        as a replayed graph) first waits for whatever ``prefetch`` still has in flight on its own stream."""
        if self._capture_keep is not None:
            return      # inside a graph capture: the caller (forward / dp_front) already waited before _graph_call; an event recorded on a
                        # non-capturing stream must not be waited on from the capture stream
        look = getattr(self, "_look_stream", None)
        cur = torch.cuda.current_stream(self.device)
        if look is not None and cur != look:
            cur.wait_stream(look)

    def project(self, tokens: torch.Tensor) -> torch.Tensor:
        """mlp1 on pre-projector tokens [..., 4*Hv] -> [..., H] (modeling_internvl_chat.py:529)."""
        lib, ctx = self._native()
        t = tokens.to(device=self.device, dtype=torch.bfloat16).contiguous()
        rows = t.numel() // t.shape[-1]
        out = torch.empty(t.shape[:-1] + (self.config.llm_config.hidden_size,), dtype=torch.bfloat16, device=self.device)
        native.check(lib.aigv_project(ctx, t.data_ptr(), rows, out.data_ptr(), native.stream_ptr()), ctx)
        return out

    def extract_feature(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """modeling_internvl_chat.py:508-531: [F,3,S,S] -> [F, num_image_token, llm_hidden]."""
        return self.project(self.vit_tokens(pixel_values))

    def motion_embed(self, motion_feature: torch.Tensor) -> torch.Tensor:
        """motion_mlp on the SlowFast feature [B, motion_dim] -> [B, H] (modeling_internvl_chat.py:344-345)."""
        b = motion_feature.shape[0]
        self._join_side_stream()
        lib, ctx = self._native(n_clips=b)
        m = motion_feature.reshape(b, -1).to(device=self.device, dtype=torch.bfloat16).contiguous()
        if m.shape[1] != self.config.motion_dim:
            raise ValueError(f"motion_feature must be [B,{self.config.motion_dim}]")
        out = torch.empty((b, self.config.llm_config.hidden_size), dtype=torch.bfloat16, device=self.device)
        native.check(lib.aigv_motion_project(ctx, m.data_ptr(), b, out.data_ptr(), native.stream_ptr()), ctx)
        return out

    def motion_feature(self, pixel_values: torch.Tensor, batch: int) -> torch.Tensor:
        """SlowFast feature [batch, motion_dim] of the clips in ``pixel_values`` [batch * T, 3, S, S] (modeling_internvl_chat.py:336-343)."""
        out = self._motion_feature(pixel_values, batch, None)
        self._join_side_stream()
        return out

    def motion_feature_async(self, pixel_values: torch.Tensor, batch: int) -> torch.Tensor:
        """The same, without joining the side stream: the tensor is only safe to consume through ``forward(motion_feature=...)`` /
        ``motion_embed``, which join it (used by the data-parallel scorer to start the branch before its ViT shard)."""
        return self._motion_feature(pixel_values, batch, None)

    def _join_side_stream(self):
        if getattr(self, "_side_pending", False):
            torch.cuda.current_stream().wait_stream(self._side_stream)
            self._side_pending = False

    def _motion_feature(self, pixel_values, batch, motion_feature):
        if motion_feature is not None:
            return motion_feature
        if self.slowfast_model is None:
            raise RuntimeError("the SlowFast motion branch is an input of this path: pass motion_feature=[B, "
                               f"{self.config.motion_dim}] or set model.slowfast_model (SURVEY.md §2 row 6)")
        if hasattr(self.slowfast_model, "features"):     # the native branch reads pixel_values as they are and samples the slow pathway itself
            pv = pixel_values.to(self.device)
            if not getattr(self, "overlap_motion_branch", True):
                return self.slowfast_model.features(pv, batch)
            # The branch depends on the frames only and its result is needed after ViT + projector: enqueue it on a side stream so that its
            # low-occupancy kernels (the slow pathway's deep layers run ~100 workgroups) fill in around the ViT's; motion_embed() joins.
            cur = torch.cuda.current_stream()
            side = getattr(self, "_side_stream", None)
            if side is None:
                side = self._side_stream = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                feat = self.slowfast_model.features(pv, batch)
            feat.record_stream(cur)
            pv.record_stream(side)
            self._side_pending = True
            return feat
        # a user-supplied callable: reference data flow (modeling_internvl_chat.py:337-344, pack_pathway_output :97-133)
        S = self.config.image_size
        frames = pixel_values.view(batch, pixel_values.shape[0] // batch, 3, S, S).permute(0, 2, 1, 3, 4)
        idx = torch.linspace(0, frames.shape[2] - 1, frames.shape[2] // 4).long().to(frames.device)
        with torch.no_grad():
            return self.slowfast_model([frames.index_select(2, idx), frames]).view(batch, -1)

    @staticmethod
    def _pack(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]):
        """Strip padding: returns (packed ids [T], cu_seqlens list, packed-row index of every [b, p] or -1)."""
        b, n = input_ids.shape
        mask = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
        lens = mask.sum(1).tolist()
        cu = [0]
        for x in lens:
            cu.append(cu[-1] + int(x))
        row_of = torch.full((b, n), -1, dtype=torch.long, device=input_ids.device)
        row_of[mask] = torch.arange(cu[-1], device=input_ids.device)
        return input_ids[mask].contiguous(), cu, row_of

    def _h2d(self, t):
        """Host tensor -> device through pinned memory without blocking the host.  While a HIP graph is being captured (``capture_forward``)
        the pinned staging buffer is kept alive with the graph: its replays copy from that very address."""
        if t.is_cuda:
            return t
        pinned = t.contiguous().pin_memory()
        keep = getattr(self, "_capture_keep", None)
        if keep is not None:
            keep.append(pinned)
        return pinned.to(self.device, non_blocking=True)

    def _prefill(self, ids_packed, slot, cu, vis, n_vis, motion, score_rows, logit_rows, keep_kv=False, kv_cap=0):
        b = len(cu) - 1
        T = cu[-1]
        lib, ctx = self._native(n_tokens=T, n_clips=b, out_rows=len(logit_rows), kv_cap=kv_cap)
        dev = self.device
        def up(t, dt):   # host index arrays go up through pinned memory without blocking the host
            return self._h2d(t.to(dt).contiguous())
        ids_d = up(ids_packed, torch.long)
        slot_d = up(slot, torch.int32)
        score = torch.empty(b, dtype=torch.float32, device=dev) if score_rows is not None else None
        amax = torch.empty(max(len(logit_rows), 1), dtype=torch.long, device=dev)
        cu_a = native.i32_array(cu)
        sr_a = native.i32_array(score_rows) if score_rows is not None else None
        lr_a = native.i32_array(logit_rows) if len(logit_rows) else None
        native.check(lib.aigv_llm_prefill(
            ctx, ids_d.data_ptr(), slot_d.data_ptr(), cu_a, b, native.ptr(vis), n_vis, native.ptr(motion),
            sr_a, native.ptr(score), lr_a, len(logit_rows), amax.data_ptr(), int(keep_kv), native.stream_ptr()), ctx)
        return score, amax[: len(logit_rows)]

    def forward(self, mos: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None,
                input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids=None, image_flags: Optional[torch.Tensor] = None, past_key_values=None,
                labels: Optional[torch.Tensor] = None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, motion_feature: Optional[torch.Tensor] = None,
                visual_tokens: Optional[torch.Tensor] = None, full_logits: bool = False):
        """Stage-2 eval pass (modeling_internvl_chat.py:306-488) or, with ``stage=1``, the stage-1 pass
        (internvl_chat_eval1/modeling_internvl_chat.py:250-366).  ``visual_tokens`` optionally supplies
        already all-gathered pre-projector tokens (frame-DP) instead of ``pixel_values``."""
        if position_ids is not None or past_key_values is not None:
            raise NotImplementedError("the eval pass takes default positions and no cache, like the reference drivers")
        if self.img_context_token_id is None:
            raise AssertionError("img_context_token_id must be set by the caller (stage2_eval.py:810)")
        pixel_values, visual_tokens, motion_feature = self._take_ahead(pixel_values, visual_tokens, motion_feature)
        if self._graph_replay_enabled and self._capture_keep is None:
            out = self._forward_through_graph(mos, pixel_values, input_ids, attention_mask, image_flags, labels, motion_feature, visual_tokens, full_logits)
            if out is not None:
                return out
        B, N = input_ids.shape
        n_frames = visual_tokens.shape[0] if visual_tokens is not None else pixel_values.shape[0]
        # ---- index bookkeeping first, on the host (one small D2H copy if the ids live on the device), so that
        # every kernel of the step can then be enqueued back to back without a host sync in between ----
        plan = self._plan(input_ids, attention_mask, labels, image_flags, n_frames, full_logits)
        motion_feature = self._motion_feature(pixel_values, B, motion_feature)

        # ---- device work: ViT -> projector -> motion projector -> LLM pass + heads ----
        self._native(n_frames=n_frames, n_tokens=plan["cu"][-1], n_clips=B, out_rows=len(plan["logit_rows"]), seq_len=N)   # size workspaces once
        vit_embeds, motion = self._visual_inputs(pixel_values, visual_tokens, motion_feature, plan)
        score, amax = self._prefill(plan["ids_packed"], plan["slot"], plan["cu"], vit_embeds, plan["n_vis"], motion,
                                    plan["score_rows"], plan["logit_rows"])
        return self._outputs(plan, B, N, score, amax, mos)

    # ---- HIP-graph replay of whole scoring passes (opt-in: enable_graph_replay) ---------------------------------------------------------
    _graph_replay_enabled = False
    _capture_keep = None
    GRAPH_CACHE_SIZE = 8
    PARKED_GRAPHS_LIMIT = 384     # dropped graphs kept alive before capturing stops for good (~95 MB each for a 4-clip pass at 8B sizes: ~36 GB)

    def enable_graph_replay(self, on: bool = True):
        """``forward`` calls whose HOST-side arguments (token ids, masks, labels, frame flags, options) and tensor shapes repeat - the
        reference's eval loop scores every clip behind the same prompt (stage2_eval.py:908-941) - are captured into a HIP graph on their
        second occurrence and replayed from the third on: ONE host call launches the ~1000 kernels of the pass (InternViT, projector,
        SlowFast side stream, InternLM2, heads), the frames are copied into the graph's input buffer first.  Same kernels, same bits
        (tests/test_gpu_api.py); what changes is the host time per pass (5-6 ms -> ~0.1 ms) - decisive where the host is slower than the
        GPU's launch stream (a CPU-throttled container: 314 -> 115 ms per step measured, profiles/r5_graph_replay.txt).  Off by default;
        any weight / mode / knob change drops the captured graphs (after a device synchronisation).  At most GRAPH_CACHE_SIZE call shapes are
        tracked; a captured graph is never evicted while the model runs - once every entry holds a graph, further call shapes stay eager (round 6:
        a 1200-clip soak with ragged groups showed that destroying a graph to make room, possibly with its replay still in flight, poisons a later
        capture: tests/manual/soak_loop.py)."""
        self._graph_replay_enabled = bool(on)
        self._drop_graphs()
        self._graphs = {}

    def _drop_graphs(self):
        """Forget every captured pass (a weight / mode / knob / capacity change made them stale).  The graph OBJECTS are not destroyed: they are parked in a
        process-wide list until the interpreter exits.  On this stack (ROCm 7.2, torch 2.10) destroying a graph object - like any device-memory release - INSIDE a
        stream capture kills that capture (the process aborts or can launch nothing any more: scripts/capture_hipfree_probe.py), an object that is merely dropped
        may be destroyed at any later moment by Python's cyclic collector, also in the middle of another capture, and destroying them at a quiet moment (device
        idle, no capture underway, followed by empty_cache) was tried and is not safe either: with several models alive, a later replay of ANOTHER model's live
        graph then crashed inside hipGraphLaunch (tests/manual/fuzz_api.py seed 3; profiles/r6_soak.txt).  Parking costs the graph's private pool and static copies
        (~95 MB per captured 4-clip pass at 8B sizes) plus ~1.5 MiB of runtime memory, at most GRAPH_CACHE_SIZE graphs per drop; drops happen on weight / mode / capacity changes and when the
        motion branch retires a native handle, i.e. rarely."""
        if getattr(self, "_graphs", None):
            held = [v for v in self._graphs.values() if isinstance(v, tuple)]
            if held and self.device.type == "cuda":
                torch.cuda.synchronize(self.device)      # (no replay in flight while the entries change hands)
            _PARKED_GRAPHS.extend(held)
            self._graphs = {}

    def _branch_uid(self):
        sf = self.slowfast_model
        return None if sf is None else getattr(sf, "uid", None) or ("id", id(sf))

    def _prepare_motion_branch(self, frames, n_clips: int):
        """In front of every pass that may be captured or replayed with the native SlowFast branch inside: make the branch's native handle for
        this geometry exist NOW (creating one allocates and uploads weights - illegal inside a capture), and drop this model's graphs when any
        handle of the branch has been destroyed since they were captured (a graph holds the addresses of a handle's buffers; SlowFastR50.epoch)."""
        sf = self.slowfast_model
        if sf is None or not hasattr(sf, "prepare"):
            return
        if frames is not None and n_clips > 0 and frames.dim() == 4 and frames.shape[0] % n_clips == 0:
            sf.prepare(self.device, int(n_clips), int(frames.shape[0]) // int(n_clips), int(frames.shape[2]), int(frames.shape[3]))
        seen = (self._branch_uid(), sf.epoch)
        if getattr(self, "_sf_epoch", None) != seen:
            if getattr(self, "_sf_epoch", None) is not None and any(isinstance(v, tuple) for v in self.__dict__.get("_graphs", {}).values()):
                self._drop_graphs()
            self._sf_epoch = seen

    def _graph_call(self, host_key, dev_inputs, fn, clone_outputs=True):
        """Graph-cached call of ``fn(*dev_inputs)`` (launches on torch's current stream only; device tensors in, a tensor / tuple / dict of
        device tensors out): first occurrence of (host_key, input shapes) -> None (the caller runs eager); second -> capture on static copies
        of the inputs; afterwards copy the inputs in, replay, hand the outputs back (cloned unless the caller consumes them at once)."""
        key = (host_key, tuple(None if t is None else (tuple(t.shape), t.dtype) for t in dev_inputs))
        graphs = self.__dict__.setdefault("_graphs", {})
        ent = graphs.get(key)
        if os.environ.get("AIGV_GRAPH_DEBUG"):
            import sys as _s
            print(f"[graph] {host_key[0]} key#{hash(key) & 0xffff:04x} state={'new' if ent is None else ent if isinstance(ent, str) else 'captured'} cache={len(graphs)} "
                  f"stream={torch.cuda.current_stream(self.device).cuda_stream:#x}", file=_s.stderr, flush=True)
        if ent is None:                      # first occurrence: eager (sizes the context, warms every kernel); remember the key
            if len(graphs) >= self.GRAPH_CACHE_SIZE:
                # make room by forgetting a key that holds no graph; a CAPTURED graph is never destroyed while the model runs (only by _drop_graphs:
                # a weight / mode / capacity change) - when all entries hold graphs, further call shapes simply stay eager
                victim = next((k for k, v in graphs.items() if isinstance(v, str)), None)
                if victim is None:
                    return None
                graphs.pop(victim)
            graphs[key] = "seen"
            return None
        if ent == "eager":
            return None
        if ent == "seen" and len(_PARKED_GRAPHS) >= self.PARKED_GRAPHS_LIMIT:
            # dropped graphs cannot be destroyed safely on this stack (_drop_graphs): they are parked, with their memory.  A process that has dropped this many
            # (hundreds of mode / weight / capacity changes under graph replay) stops capturing instead of running out of device memory: same kernels, same bits, eager
            if not getattr(InternVLChatModel, "_park_limit_warned", False):
                InternVLChatModel._park_limit_warned = True
                import warnings
                warnings.warn(f"graph replay: {len(_PARKED_GRAPHS)} dropped graphs are parked (they cannot be destroyed safely on this ROCm build); no further pass is "
                              "captured in this process - the eager path runs the same kernels")
            graphs[key] = "eager"
            return None
        if ent == "seen":                    # second occurrence: capture, on static copies of the device inputs
            statics = [None if t is None else t.clone() for t in dev_inputs]
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            self._capture_keep = []
            try:
                with native.capturing(), torch.cuda.graph(graph, capture_error_mode="relaxed"):
                    outputs = fn(*statics)
                keep = self._capture_keep
            except Exception as e:           # a pass that does not capture (an allocation or a synchronisation inside it) stays eager for good - and
                import warnings              # says so; the eager run that follows raises whatever was a real error rather than a capture-illegal call
                import traceback
                where = "".join(traceback.format_tb(e.__traceback__)[-3:])
                warnings.warn(f"graph replay: capture of {host_key[0]!r} failed ({type(e).__name__}: {str(e).splitlines()[0]}); this call shape stays eager\n{where}")
                graphs[key] = "eager"
                self._capture_keep = None
                native.load().aigv_clear_hip_error()
                try:                             # can this process still launch?  (scripts/capture_error_probe.py: on ROCm 7.2 a capture that an illegal call INVALIDATED
                    torch.zeros(1, device=self.device).add_(1)      # is never ended - hipStreamEndCapture on it crashes - and every later launch on any stream
                    torch.cuda.synchronize(self.device)             # fails with hipErrorStreamCaptureInvalidated: there is nothing to fall back to)
                except Exception as dead:
                    raise native.NativeError(
                        f"HIP-graph capture of {host_key[0]!r} was invalidated ({type(e).__name__}: {str(e).splitlines()[0]}) and this ROCm build cannot recover from that: every "
                        "further kernel launch of the process fails.  Restart without enable_graph_replay() - the eager path runs the same kernels - and report the call "
                        "sequence that led here") from dead
                return None
            finally:
                self._capture_keep = None
            ent = graphs[key] = (graph, outputs, statics, keep)
        graph, outputs, statics, _keep = ent
        for st, t in zip(statics, dev_inputs):
            if st is not None:
                st.copy_(t)
        graph.replay()
        if not clone_outputs:
            return outputs
        cl = lambda v: v.clone() if torch.is_tensor(v) else v      # (the graph's own output tensors are overwritten by the next replay)
        if isinstance(outputs, dict):
            return {k: cl(v) for k, v in outputs.items()}
        if isinstance(outputs, (tuple, list)):
            return tuple(cl(v) for v in outputs)
        return cl(outputs)

    def _forward_through_graph(self, mos, pixel_values, input_ids, attention_mask, image_flags, labels, motion_feature, visual_tokens, full_logits):
        """The replay path of ``forward``; returns None when the call does not qualify (the eager path then runs)."""
        src = visual_tokens if visual_tokens is not None else pixel_values
        if self._rope_seq_len(int(input_ids.shape[1])) != getattr(self, "_rope_ntk", 0):
            return None      # this pass re-derives the rotary tables (dynamic NTK: another sequence length than the last pass) - synchronous uploads, never inside a capture
        if (mos is not None or src is None or not src.is_cuda or self._dirty or self._ctx is None or getattr(self, "_prof_on", False)
                or (motion_feature is not None and not motion_feature.is_cuda) or (visual_tokens is not None and motion_feature is None)):
            return None
        host = lambda t: None if t is None else t.detach().to("cpu").contiguous()
        parts = [host(input_ids), host(attention_mask), host(labels), host(image_flags)]
        host_key = ("forward", visual_tokens is not None, bool(full_logits), int(self.img_context_token_id), self._branch_uid(),
                    bool(getattr(self, "overlap_motion_branch", True)), bool(getattr(self, "drop_dead_tail", True)),
                    tuple(None if t is None else (tuple(t.shape), t.dtype, t.numpy().tobytes()) for t in parts))
        self._join_side_stream()             # (a motion feature started by motion_feature_async: joined BEFORE the graph copies it in)
        self._prepare_motion_branch(pixel_values if (motion_feature is None and visual_tokens is None) else None, int(input_ids.shape[0]))

        def fn(src_static, mf_static):
            return self.forward(mos=None, pixel_values=None if visual_tokens is not None else src_static, input_ids=input_ids, attention_mask=attention_mask,
                                image_flags=image_flags, labels=labels, motion_feature=mf_static, full_logits=full_logits,
                                visual_tokens=src_static if visual_tokens is not None else None)
        return self._graph_call(host_key, [src, motion_feature], fn)

    def dp_front(self, frames_local: torch.Tensor, frames_clips: Optional[torch.Tensor], n_clips: int):
        """The data-parallel scorer's front half on this rank (dist_utils.score_clips_dp): the SlowFast feature of its own clips (side
        stream) beside the InternViT tokens of its frame shard -> (tokens [F_local, ntok, 4 Hv], motion feature [n_clips, motion_dim] or
        None).  With graph replay enabled the two run as ONE captured graph (joined at its end); the token all-gather and the projector +
        InternLM2 half (``forward(visual_tokens=...)``, a graph of its own) follow on the host's side of the collective."""
        self._wait_for_prefetch()

        def fn(fl, fc):
            mf = self._motion_feature(fc, n_clips, None) if fc is not None else None
            tok = self.vit_tokens(fl)
            self._join_side_stream()
            return tok, mf
        if (self._graph_replay_enabled and self._capture_keep is None and frames_local.is_cuda and not self._dirty and self._ctx is not None
                and not getattr(self, "_prof_on", False) and (frames_clips is None or (frames_clips.is_cuda and hasattr(self.slowfast_model, "features")))):
            self._prepare_motion_branch(frames_clips, int(n_clips))
            out = self._graph_call(("dp_front", int(n_clips), self._branch_uid(), bool(getattr(self, "overlap_motion_branch", True))),
                                   [frames_local, frames_clips], fn, clone_outputs=False)
            if out is not None:
                return out
        if frames_clips is None:
            return self.vit_tokens(frames_local), None
        mf = self.motion_feature_async(frames_clips, n_clips)       # eager: joined where forward() consumes it
        return self.vit_tokens(frames_local), mf

    def prefetch(self, pixel_values: Optional[torch.Tensor] = None, frames_u8: Optional[torch.Tensor] = None, n_clips: int = 1) -> VisualAhead:
        """Start the visual front of a LATER ``forward`` call NOW, on a stream of its own: frame ingest (when ``frames_u8`` [F, H, W, 3] is
        given: H2D copy + Pillow-exact resize + normalise), InternViT + pixel-shuffle, and the SlowFast branch of the ``n_clips`` clips -
        everything that depends on the frames only.  The returned handle is passed to ``forward`` as ``pixel_values``; that call waits for
        the handle's event and runs projector + InternLM2 + heads.  In an eval loop that scores one clip per call (stage2_eval.py:908-941)
        the next clip's visual front then runs BESIDE the current clip's InternLM2 pass, whose wo / w2 launches leave half the CUs idle at
        one clip (``eval_utils.lookahead`` wraps a loop that way).  Same kernels, same bits as the plain call; the InternViT workspaces of
        the context serve one visual front at a time, so a prefetch waits for the previous one.  (HIP deals a process's streams round-robin onto a
        few hardware queues: should the prefetch stream land on the queue of the caller's stream, the two serialise and the loop runs at the plain
        loop's speed - with the same results.)"""
        if (pixel_values is None) == (frames_u8 is None):
            raise ValueError("prefetch takes pixel_values or frames_u8")
        cur = torch.cuda.current_stream(self.device)
        look = getattr(self, "_look_stream", None)
        if look is None:
            look = self._look_stream = torch.cuda.Stream(device=self.device)
        look.wait_stream(cur)                       # inputs produced on the caller's stream; the previous prefetch is ordered by the stream itself
        with torch.cuda.stream(look):
            pv = self.ingest_frames(frames_u8) if frames_u8 is not None else pixel_values.to(device=self.device, dtype=torch.bfloat16)
            need_motion = self.slowfast_model is not None and hasattr(self.slowfast_model, "features")
            tok, mf = self.dp_front(pv, pv if need_motion else None, n_clips)
            self._join_side_stream()                # (the eager SlowFast branch forks from and joins back into this stream)
            tok = tok.clone()                       # (a replayed graph hands out its own output buffers: the next prefetch overwrites them)
            mf = None if mf is None else mf.clone()
            ev = torch.cuda.Event()
            ev.record(look)
        for t in ([pixel_values] + (list(frames_u8) if isinstance(frames_u8, (list, tuple)) else [frames_u8])):
            if t is not None and t.is_cuda:
                t.record_stream(look)
        tok.record_stream(cur)
        if mf is not None:
            mf.record_stream(cur)
        return VisualAhead(tok, mf, ev, n_clips)

    def _plan(self, input_ids, attention_mask, labels, image_flags, n_frames, full_logits=False, drop_dead_tail=None):
        """Host-side token bookkeeping of one pass: packed ids, which packed row takes which visual / motion token
        (modeling_internvl_chat.py:351-378), and the rows whose outputs are consumed."""
        B, N = input_ids.shape
        ids_h = input_ids.detach().to("cpu")
        mask_h = attention_mask.detach().to("cpu") if attention_mask is not None else None
        labels_h = labels.detach().to("cpu") if labels is not None else torch.full_like(ids_h, -100)
        flags_h = image_flags.detach().to("cpu").squeeze(-1) if image_flags is not None else None
        ids_packed, cu, row_of = self._pack(ids_h, mask_h)
        lens = [cu[i + 1] - cu[i] for i in range(B)]
        sel = ids_packed == self.img_context_token_id
        seq_of = torch.repeat_interleave(torch.arange(B), torch.tensor(lens))
        # last <IMG_CONTEXT> of each clip <- motion token; the others, in order <- visual tokens (:351-378)
        pos_idx = torch.arange(ids_packed.numel())
        last_pos = torch.full((B,), -1, dtype=torch.long)
        last_pos.scatter_reduce_(0, seq_of[sel], pos_idx[sel], reduce="amax")
        if bool((last_pos < 0).any()):
            raise ValueError("every clip needs at least one <IMG_CONTEXT> token")
        is_motion = torch.zeros_like(sel)
        is_motion[last_pos] = True
        vis_sel = sel & ~is_motion
        keep = torch.arange(n_frames) if flags_h is None else (flags_h == 1).nonzero().flatten()
        n_vis = int(keep.numel()) * self.num_image_token
        if int(vis_sel.sum()) != n_vis:
            raise ValueError(f"visual token count mismatch: {int(vis_sel.sum())} <IMG_CONTEXT> slots vs {n_vis} visual tokens")
        slot = torch.full_like(ids_packed, -1, dtype=torch.int32)
        slot[vis_sel] = torch.arange(n_vis, dtype=torch.int32)
        slot[is_motion] = n_vis + seq_of[is_motion].to(torch.int32)
        # rows whose next-token argmax is consumed: shifted positions p with labels[p+1] != -100
        if full_logits:
            want = row_of[:, :-1] >= 0
        else:
            want = (labels_h[:, 1:] != -100) & (row_of[:, :-1] >= 0)
        logit_rows = row_of[:, :-1][want].tolist()
        score_rows = [cu[i + 1] - 4 for i in range(B)] if self.stage == 2 else None
        if score_rows is not None and min(lens) < 4:
            raise ValueError("clips need at least 4 tokens for the score row hidden[:, -4]")
        # Dead trailing tokens: with causal attention a token influences only later rows, so whatever follows a clip's last
        # consumed row (the closing <|im_end|>, whose own logits the reference drops with shift_logits = logits[:, :-1],
        # modeling_internvl_chat.py:451-455) changes no returned value.  Such text tokens are not run at all - for the
        # canonical clip 2177 -> 2176 = 17 x 128 rows, which also removes the ragged row tile / query block of every kernel.
        if drop_dead_tail is None:
            drop_dead_tail = getattr(self, "drop_dead_tail", True)
        if drop_dead_tail and not full_logits:
            last_needed = [cu[b] for b in range(B)]
            for r in logit_rows + (score_rows or []):
                b = int(seq_of[r])
                last_needed[b] = max(last_needed[b], r + 1)
            keep_row = torch.zeros(ids_packed.numel(), dtype=torch.bool)
            for b in range(B):
                end = max(last_needed[b], cu[b] + 1)
                if bool((slot[end:cu[b + 1]] >= 0).any()):      # never drop a visual / motion slot
                    end = cu[b + 1]
                keep_row[cu[b]:end] = True
            if not bool(keep_row.all()):
                new_index = torch.cumsum(keep_row.long(), 0) - 1
                remap = lambda rows: [int(new_index[r]) for r in rows]
                logit_rows = remap(logit_rows)
                score_rows = remap(score_rows) if score_rows is not None else None
                last_pos = new_index[last_pos]
                kept = keep_row.nonzero().flatten()
                row_of = torch.where(row_of >= 0, torch.where(keep_row[row_of.clamp_min(0)], new_index[row_of.clamp_min(0)], torch.full_like(row_of, -1)), row_of)
                ids_packed, slot, seq_of = ids_packed[kept], slot[kept], seq_of[kept]
                lens = [int((seq_of == b).sum()) for b in range(B)]
                cu = [0]
                for x in lens:
                    cu.append(cu[-1] + x)
        return dict(ids_h=ids_h, mask_h=mask_h, labels_h=labels_h, flags_h=flags_h, ids_packed=ids_packed, cu=cu, row_of=row_of,
                    lens=lens, slot=slot, n_vis=n_vis, keep=keep, n_frames=n_frames, want=want, logit_rows=logit_rows,
                    score_rows=score_rows, last_ctx=(last_pos - torch.tensor(cu[:-1])).tolist())

    def _visual_inputs(self, pixel_values, visual_tokens, motion_feature, plan):
        H = self.config.llm_config.hidden_size
        if visual_tokens is None:
            visual_tokens = self.vit_tokens(pixel_values)
        vit_embeds = self.project(visual_tokens)                       # [F, ntok, H]
        if plan["flags_h"] is not None and int(plan["keep"].numel()) != plan["n_frames"]:
            vit_embeds = vit_embeds[self._h2d(plan["keep"])]
        return vit_embeds.reshape(-1, H), self.motion_embed(motion_feature)

    def _outputs(self, plan, B, N, score, amax, mos):
        dev = self.device
        up = self._h2d   # host -> device through pinned memory, never blocking the host (keeps the CPU ahead of the GPU)
        logit = torch.full((B * (N - 1),), -1, dtype=torch.long, device=dev)
        if len(plan["logit_rows"]):
            logit.index_copy_(0, up(plan["want"].reshape(-1).nonzero().flatten()), amax)   # index list built on the host: no sync
        out = {"label": up(plan["labels_h"][..., 1:].contiguous().view(-1)), "logit": logit.view(-1)}
        if self.stage == 2:
            score1 = score.to(torch.bfloat16)       # the head computes in bf16; the value is exact in fp32
            out["score1"] = score1
            out["loss"] = F.l1_loss(score1, mos.to(dev).to(score1.dtype)) if mos is not None else None
        return out

    @staticmethod
    def _shared_prefix_lengths(plans, B: int) -> List[int]:
        """Per clip: the number of leading tokens every prompt shares, capped so that every consumed row (answer rows, score
        row) stays in the continuation; raises if the prompts diverge before the last <IMG_CONTEXT> token (host logic only)."""
        pre = []
        for b in range(B):
            seqs = [pl["ids_packed"][pl["cu"][b]:pl["cu"][b + 1]] for pl in plans]
            n = min(len(x) for x in seqs)
            eq = torch.ones(n, dtype=torch.bool)
            for x in seqs[1:]:
                eq &= x[:n] == seqs[0][:n]
            lcp = int(n if bool(eq.all()) else eq.long().argmin())
            first_needed = []
            for pl in plans:
                rows = [r - pl["cu"][b] for r in pl["logit_rows"] if pl["cu"][b] <= r < pl["cu"][b + 1]]
                if pl["score_rows"] is not None:
                    rows.append(pl["score_rows"][b] - pl["cu"][b])
                first_needed.append(min(rows) if rows else pl["lens"][b] - 1)
            p_b = min([lcp] + first_needed + [pl["lens"][b] - 1 for pl in plans])
            if p_b <= max(pl["last_ctx"][b] for pl in plans):
                raise ValueError(f"clip {b}: the prompts diverge before the last <IMG_CONTEXT> token - no shared video prefix")
            pre.append(p_b)
        return pre

    def forward_shared_prefix(self, prompts, pixel_values: Optional[torch.Tensor] = None, image_flags: Optional[torch.Tensor] = None,
                              motion_feature: Optional[torch.Tensor] = None, visual_tokens: Optional[torch.Tensor] = None, mos=None):
        """Score the same clips under several prompts that share their beginning - the reference's four quality
        perspectives ask four questions BEHIND the same system + frame + motion tokens (8f-3) and
        run four full passes (stage2_eval.py evaluates one jsonl per perspective).  Here the common prefix runs once
        (ViT, projector, LLM prefill into the KV cache); every prompt then only continues its own few question / answer
        tokens over the cached keys (``aigv_llm_extend``).  ``prompts``: list of ``(input_ids[B, N_p], attention_mask,
        labels)``; returns the list of ``forward`` result dicts, one per prompt.  Causal attention makes the prefix rows
        independent of what follows, so each result is that of a separate ``forward`` call up to kernel summation order."""
        if self.img_context_token_id is None:
            raise AssertionError("img_context_token_id must be set by the caller (stage2_eval.py:810)")
        if not prompts:
            return []
        pixel_values, visual_tokens, motion_feature = self._take_ahead(pixel_values, visual_tokens, motion_feature)
        n_frames = visual_tokens.shape[0] if visual_tokens is not None else pixel_values.shape[0]
        plans = [self._plan(ids, am, lab, image_flags, n_frames) for (ids, am, lab) in prompts]
        B = prompts[0][0].shape[0]
        pre = self._shared_prefix_lengths(plans, B)
        p0 = plans[0]
        ids_prefix = torch.cat([p0["ids_packed"][p0["cu"][b]:p0["cu"][b] + pre[b]] for b in range(B)])
        slot_prefix = torch.cat([p0["slot"][p0["cu"][b]:p0["cu"][b] + pre[b]] for b in range(B)])
        cu_prefix = [0]
        for b in range(B):
            cu_prefix.append(cu_prefix[-1] + pre[b])
        motion_feature = self._motion_feature(pixel_values, B, motion_feature)
        longest = max(max(pl["lens"]) for pl in plans)
        P = len(plans)
        n_suffix = sum(pl["cu"][-1] for pl in plans) - P * cu_prefix[-1]
        self._native(n_frames=n_frames, n_tokens=max(cu_prefix[-1], n_suffix), n_clips=B * P,
                     out_rows=sum(len(pl["logit_rows"]) for pl in plans), kv_cap=longest + 1,
                     seq_len=max(ids.shape[1] for (ids, _, _) in prompts))
        vit_embeds, motion = self._visual_inputs(pixel_values, visual_tokens, motion_feature, p0)
        self._prefill(ids_prefix, slot_prefix, cu_prefix, vit_embeds, p0["n_vis"], motion, None, [], keep_kv=True, kv_cap=longest + 1)
        lib, ctx = native.load(), self._ctx
        dev = self.device
        # one cache copy per prompt, then ONE continuation pass over B * P sequences (sequence p * B + b = clip b under prompt p):
        # the decoder weights are streamed once for all prompts
        native.check(lib.aigv_kv_fork(ctx, P, native.stream_ptr()), ctx)
        parts, cu_s, lrows, srows, n_l = [], [0], [], [], []
        for pl in plans:
            starts = []
            for b in range(B):
                parts.append(pl["ids_packed"][pl["cu"][b] + pre[b]:pl["cu"][b + 1]])
                starts.append(cu_s[-1])
                cu_s.append(cu_s[-1] + pl["lens"][b] - pre[b])
            def local(r, pl=pl, starts=starts):   # packed row of the full prompt -> packed row of the suffix batch
                b = max(i for i in range(B) if pl["cu"][i] <= r)
                return starts[b] + (r - pl["cu"][b] - pre[b])
            rows = [local(r) for r in pl["logit_rows"]]
            lrows += rows
            n_l.append(len(rows))
            if pl["score_rows"] is not None:
                srows += [local(r) for r in pl["score_rows"]]
        ids_d = torch.cat(parts).to(torch.long).contiguous().pin_memory().to(dev, non_blocking=True)
        score = torch.empty(B * P, dtype=torch.float32, device=dev) if self.stage == 2 else None
        amax = torch.empty(max(len(lrows), 1), dtype=torch.long, device=dev)
        native.check(lib.aigv_llm_extend(ctx, ids_d.data_ptr(), native.i32_array(cu_s), B * P,
                                         native.i32_array(srows) if score is not None else None, native.ptr(score),
                                         native.i32_array(lrows) if lrows else None, len(lrows), amax.data_ptr(), 0,
                                         native.stream_ptr()), ctx)
        outs, off = [], 0
        for p, (pl, (ids, _, _)) in enumerate(zip(plans, prompts)):
            outs.append(self._outputs(pl, B, ids.shape[1], score[p * B:(p + 1) * B] if score is not None else None,
                                      amax[off:off + n_l[p]], mos))
            off += n_l[p]
        return outs

    # ---- generation (API surface; greedy) -------------------------------------------------------------------
    EOS_CHECK_EVERY = 8     # tokens between two host reads of the device-side "finished" flags

    def _greedy(self, ids_packed, slot, cu, vis, n_vis, max_new_tokens: int, eos_ids: List[int], pad_id: int, motion=None, sampler=None,
                processors=None, beams=None):
        """The token loop of generate(): HF's greedy search / multinomial sampling loop (the reference calls ``language_model.generate``,
        modeling_internvl_chat.py:798-809).  The end-of-sequence bookkeeping runs on the device (aigv_decode_eos): a finished sequence
        emits ``pad_id``, the loop stops once every sequence has emitted an end token - checked by the host only every EOS_CHECK_EVERY
        tokens, so no per-token host synchronisation; the columns past HF's stopping point are cut off afterwards."""
        b = len(cu) - 1
        longest = max(cu[i + 1] - cu[i] for i in range(b))
        last_rows = [cu[i + 1] - 1 for i in range(b)]
        nb = beams["num_beams"] if beams else 1
        self._native(seq_len=longest, n_clips=b * nb, out_rows=b * nb)       # (beam search: room for every beam before the prompt pass)
        _, nxt = self._prefill(ids_packed, slot, cu, vis, n_vis, motion, None, last_rows, keep_kv=True,
                               kv_cap=longest + max_new_tokens + 1)
        lib, ctx = native.load(), self._ctx
        if beams:
            return self._beam_decode(b, [cu[i + 1] - cu[i] for i in range(b)], max_new_tokens, eos_ids, pad_id, processors or [], **beams)
        ntk_decode = self._rope_seq_len(longest + max_new_tokens) != 0
        eos_a = (C.c_int64 * max(len(eos_ids), 1))(*[int(e) for e in eos_ids]) if eos_ids else None
        state = torch.zeros(b + 1, dtype=torch.int32, device=self.device)     # finished flags + live-column count (aigv_amd.h)
        # aigv_decode_eos takes at most 8 end ids (kernel-argument array): longer lists keep HF's bookkeeping in torch ops on the device -
        # the same rule (next = next * unfinished + pad * (1 - unfinished); unfinished &= next not in eos), still without a per-token sync
        host_eos = len(eos_ids) > 8
        eos_t = torch.tensor([int(e) for e in eos_ids], dtype=torch.long, device=self.device) if host_eos else None
        outs: List[torch.Tensor] = []

        def eos_step(tok):
            live = state[:b] == 0
            tok = torch.where(live, tok, torch.full_like(tok, int(pad_id)))
            state[b] += live.any().to(torch.int32)
            state[:b] |= (live & torch.isin(tok, eos_t)).to(torch.int32)
            return tok

        def pick(greedy_tok):
            """The step's raw token: the fused argmax, or - with logits processors / sampling - a choice over the rows' lm-head logits."""
            if sampler is None and not processors:
                return greedy_tok
            logits = self._row_logits(b)
            if processors:
                hist = torch.stack(outs, dim=1) if outs else torch.zeros((b, 0), dtype=torch.long, device=self.device)
                for proc in processors:
                    logits = proc(hist, logits)
            return self._sample(logits, **sampler) if sampler is not None else logits.argmax(-1)

        tok = pick(nxt).contiguous()
        for step in range(max_new_tokens):
            if host_eos:
                tok = eos_step(tok).contiguous()
            elif eos_ids:     # tok: raw -> emitted (pad for finished sequences); flags / live-column count advance on the device
                native.check(lib.aigv_decode_eos(ctx, tok.data_ptr(), state.data_ptr(), eos_a, len(eos_ids), int(pad_id), native.stream_ptr()), ctx)
            outs.append(tok)
            if step + 1 == max_new_tokens:
                break
            if eos_ids and (step + 1) % self.EOS_CHECK_EVERY == 0 and bool(state[:b].all()):
                break
            new = torch.empty_like(tok)
            if ntk_decode:
                self._rope_for_decode(longest + step + 1)
            native.check(lib.aigv_decode_step(ctx, tok.data_ptr(), new.data_ptr(), native.stream_ptr()), ctx)
            tok = pick(new).contiguous()
        out = torch.stack(outs, dim=1)
        if eos_ids:
            out = out[:, : max(1, int(state[b].item()))]     # HF stops after the column in which the last live sequence ended
        return out

    def _beam_decode(self, b: int, prompt_lens: List[int], max_new_tokens: int, eos_ids: List[int], pad_id, processors, num_beams: int,
                     length_penalty: float = 1.0, early_stopping=False) -> torch.Tensor:
        """HF beam search (beam.beam_search) behind a prompt pass that kept its KV: the prompts' caches are replicated once per beam
        (aigv_kv_fork: sequence k * b + i is beam k of prompt i), every step decodes all b * num_beams sequences in one aigv_decode_step
        (the decoder weights stream once for all beams) and the chosen parents are gathered in the cache (aigv_kv_reorder)."""
        from . import beam
        lib, ctx = native.load(), self._ctx
        V = self.config.llm_config.vocab_size
        n = b * num_beams
        first = self._row_logits(b)
        native.check(lib.aigv_kv_fork(ctx, num_beams, native.stream_ptr()), ctx)
        fed = [0]                                  # tokens every beam has been fed so far = cached positions behind its prompt
        slot_of = lambda i, k: k * b + i

        def reorder(parent: torch.Tensor):
            if fed[0] == 0:
                return                             # the copies are still identical
            src = beam.parents_to_slots(parent.cpu(), slot_of)
            if src == list(range(n)):
                return
            lens = [prompt_lens[s % b] + fed[0] for s in range(n)]
            native.check(lib.aigv_kv_reorder(ctx, native.i32_array(src), native.i32_array(lens), n, native.stream_ptr()), ctx)

        ntk_decode = self._rope_seq_len(max(prompt_lens) + max_new_tokens) != 0

        def step(tok: torch.Tensor) -> torch.Tensor:
            t = tok.t().contiguous().view(-1)      # [b, nb] -> cache order
            new = torch.empty_like(t)
            if ntk_decode:
                self._rope_for_decode(max(prompt_lens) + fed[0] + 1)
            native.check(lib.aigv_decode_step(ctx, t.data_ptr(), new.data_ptr(), native.stream_ptr()), ctx)
            fed[0] += 1
            return self._row_logits(n).view(num_beams, b, V).transpose(0, 1)

        return beam.beam_search(first, step, reorder, num_beams, max_new_tokens, eos_ids=eos_ids, pad_id=pad_id, length_penalty=length_penalty,
                                early_stopping=early_stopping, processors=processors)

    @staticmethod
    def _gen_args(generation_config, kw):
        """(max_new_tokens, eos ids, pad id, sampler, processors, beams) from a HF-style generation config / kwargs.  ``sampler`` is None for greedy
        decoding or the warper settings of HF's multinomial sampling (temperature -> top-k -> top-p, transformers' order and
        defaults: top_k 50, top_p 1.0, temperature 1.0); ``processors`` = HF's repetition-penalty / no-repeat-n-gram logits processors
        when asked for; ``beams`` is None or HF's beam-search settings (num_beams > 1: num_beams, length_penalty, early_stopping)."""
        cfg = dict(generation_config) if isinstance(generation_config, dict) else {}
        if generation_config is not None and not isinstance(generation_config, dict):
            cfg = {k: getattr(generation_config, k) for k in ("max_new_tokens", "do_sample", "num_beams", "eos_token_id", "pad_token_id",
                                                              "temperature", "top_k", "top_p", "repetition_penalty", "no_repeat_ngram_size",
                                                              "length_penalty", "early_stopping", "num_return_sequences", "num_beam_groups")
                   if hasattr(generation_config, k)}
        cfg.update(kw)
        beams = None
        if (cfg.get("num_beams") or 1) > 1:
            if cfg.get("do_sample"):
                raise NotImplementedError("beam-search multinomial sampling is not implemented on the gfx950 path (beam search, greedy and sampling are)")
            if (cfg.get("num_return_sequences") or 1) != 1 or (cfg.get("num_beam_groups") or 1) != 1:
                raise NotImplementedError("beam search returns the best hypothesis only (num_return_sequences = 1, no beam groups)")
            beams = dict(num_beams=int(cfg["num_beams"]), length_penalty=float(cfg["length_penalty"]) if cfg.get("length_penalty") is not None else 1.0,
                         early_stopping=cfg.get("early_stopping") if cfg.get("early_stopping") is not None else False)
        processors = []       # HF's order (GenerationMixin._get_logits_processor): repetition penalty, then n-gram blocking
        if cfg.get("repetition_penalty") not in (None, 1, 1.0):
            processors.append(InternVLChatModel._repetition_penalty(float(cfg["repetition_penalty"])))
        if cfg.get("no_repeat_ngram_size") not in (None, 0):
            processors.append(InternVLChatModel._no_repeat_ngram(int(cfg["no_repeat_ngram_size"])))
        sampler = None
        if cfg.get("do_sample"):
            sampler = dict(temperature=float(cfg["temperature"]) if cfg.get("temperature") is not None else 1.0,
                           top_k=int(cfg["top_k"]) if cfg.get("top_k") is not None else 50,
                           top_p=float(cfg["top_p"]) if cfg.get("top_p") is not None else 1.0, generator=cfg.get("generator"))
            if sampler["temperature"] <= 0 or not (0 < sampler["top_p"] <= 1.0) or sampler["top_k"] < 0:
                raise ValueError(f"bad sampling settings {sampler}")
        eos = cfg.get("eos_token_id")
        eos = [] if eos is None else ([int(eos)] if not isinstance(eos, (list, tuple)) else [int(e) for e in eos])
        return int(cfg.get("max_new_tokens") or 20), eos, cfg.get("pad_token_id"), sampler, processors, beams

    @staticmethod
    def _repetition_penalty(penalty: float):
        """HF RepetitionPenaltyLogitsProcessor over the GENERATED tokens (the reference's generate() passes inputs_embeds, so HF's
        input_ids start empty): the logit of every token already emitted is divided by ``penalty`` if positive, multiplied if negative."""
        if penalty <= 0:
            raise ValueError("repetition_penalty must be a strictly positive float")

        def proc(hist: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
            if hist.shape[1] == 0:
                return logits
            sc = logits.gather(1, hist)
            sc = torch.where(sc < 0, sc * penalty, sc / penalty)
            return logits.scatter(1, hist, sc)
        return proc

    @staticmethod
    def _no_repeat_ngram(n: int):
        """HF NoRepeatNGramLogitsProcessor: a token that would complete an n-gram already present in the generated tokens gets -inf."""
        if n <= 0:
            raise ValueError("no_repeat_ngram_size must be a strictly positive integer")

        def proc(hist: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
            cur = hist.shape[1]
            if cur + 1 < n:
                return logits
            rows = hist.tolist()          # host glue of generate(): a few dozen tokens per sequence
            logits = logits.clone()
            for b, seq in enumerate(rows):
                prefix = tuple(seq[cur + 1 - n:cur])
                banned = [seq[i + n - 1] for i in range(cur - n + 1) if tuple(seq[i:i + n - 1]) == prefix]
                if banned:
                    logits[b, banned] = float("-inf")
            return logits
        return proc

    def _row_logits(self, n_rows: int) -> torch.Tensor:
        """fp32 [n_rows, vocab]: lm-head logits of the rows the last native pass consumed (aigv_out_row_logits) - the reference's
        ``logits = output(h).float()`` (modeling_internlm2.py:1095-1096)."""
        lib, ctx = native.load(), self._ctx
        V = self.config.llm_config.vocab_size
        ldo = (V + 3) // 4 * 4
        buf = torch.empty((n_rows, ldo), dtype=torch.bfloat16, device=self.device)
        native.check(lib.aigv_out_row_logits(ctx, 0, n_rows, buf.data_ptr(), ldo, native.stream_ptr()), ctx)
        return buf[:, :V].float()

    def last_hidden_rows(self, n_rows: int, first_row: int = 0) -> torch.Tensor:
        """bf16 [n_rows, H]: final hidden states (after the last RMSNorm) of the rows the last native pass consumed, in the order
        [score rows | logit rows] (aigv_out_row_hidden).  After ``forward`` rows 0..B-1 are the reference's
        ``hidden_states[-1][:, -4, :]`` - the score head's input (modeling_internvl_chat.py:469-481)."""
        lib, ctx = native.load(), self._ctx
        if ctx is None:
            raise native.NativeError("no native pass has run yet")
        H = self.config.llm_config.hidden_size
        buf = torch.empty((n_rows, H), dtype=torch.bfloat16, device=self.device)
        native.check(lib.aigv_out_row_hidden(ctx, first_row, n_rows, buf.data_ptr(), H, native.stream_ptr()), ctx)
        return buf

    @staticmethod
    def _sample(logits: torch.Tensor, temperature: float, top_k: int, top_p: float, generator=None) -> torch.Tensor:
        """One multinomial draw per row after HF's logits warpers in HF's order (TemperatureLogitsWarper, TopKLogitsWarper,
        TopPLogitsWarper with min_tokens_to_keep = 1; transformers/generation/logits_process.py).  Host-side glue of generate():
        a handful of torch ops on [B, vocab], not part of the scoring hot path."""
        return torch.multinomial(InternVLChatModel._warp(logits, temperature, top_k, top_p).softmax(-1), 1, generator=generator).squeeze(1)

    @staticmethod
    def _warp(logits: torch.Tensor, temperature: float, top_k: int, top_p: float) -> torch.Tensor:
        """HF's three logits warpers in HF's order (pinned against transformers' own classes in tests/test_host.py)."""
        x = logits / temperature if temperature != 1.0 else logits
        if top_k > 0:
            kth = torch.topk(x, min(top_k, x.shape[-1]))[0][..., -1, None]
            x = x.masked_fill(x < kth, float("-inf"))
        if top_p < 1.0:
            srt, idx = torch.sort(x, descending=False)
            remove = srt.softmax(-1).cumsum(-1) <= (1.0 - top_p)
            remove[..., -1:] = False
            x = x.masked_fill(remove.scatter(1, idx, remove), float("-inf"))
        return x

    @torch.no_grad()
    def generate(self, pixel_values: Optional[torch.Tensor] = None, input_ids: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, visual_features: Optional[torch.Tensor] = None,
                 generation_config=None, output_hidden_states=None, return_dict=None, **generate_kwargs) -> torch.Tensor:
        """modeling_internvl_chat.py:769-811: every <IMG_CONTEXT> slot takes a visual token (no motion
        token), then greedy decode with a KV cache.  Returns the NEW tokens [B, <=max_new_tokens]."""
        assert self.img_context_token_id is not None
        max_new, eos, pad, sampler, procs, beams = self._gen_args(generation_config, generate_kwargs)
        pad = self.config.llm_config.pad_token_id if pad is None else pad
        dev = self.device
        input_ids = input_ids.to(dev)
        ids_packed, cu, _ = self._pack(input_ids, attention_mask.to(dev) if attention_mask is not None else None)
        slot = torch.full_like(ids_packed, -1, dtype=torch.int32)
        vis, n_vis = None, 0
        if pixel_values is not None or visual_features is not None:
            vit = visual_features if visual_features is not None else self.extract_feature(pixel_values)
            vis = vit.reshape(-1, vit.shape[-1]).to(dev).contiguous()
            n_vis = vis.shape[0]
            sel = ids_packed == self.img_context_token_id
            assert int(sel.sum()) != 0
            if int(sel.sum()) != n_vis:
                raise ValueError(f"visual token count mismatch: {int(sel.sum())} slots vs {n_vis} tokens")
            slot[sel] = torch.arange(n_vis, device=dev, dtype=torch.int32)
        return self._greedy(ids_packed, slot, cu, vis, n_vis, max_new, eos, pad, sampler=sampler, processors=procs, beams=beams)

    @torch.no_grad()
    def generate2(self, input_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, visual_features=None,
                  generation_config=None, output_hidden_states=None, return_dict=None, **generate_kwargs) -> torch.Tensor:
        """modeling_internvl_chat.py:812-853: decode from precomputed input embeddings [B, N, C]."""
        max_new, eos, pad, sampler, procs, beams = self._gen_args(generation_config, generate_kwargs)
        pad = self.config.llm_config.pad_token_id if pad is None else pad
        dev = self.device
        b, n, _ = input_embeds.shape
        mask = torch.ones((b, n), dtype=torch.bool, device=dev) if attention_mask is None else attention_mask.to(dev).bool()
        emb = input_embeds.to(dev)[mask].to(torch.bfloat16).contiguous()
        lens = mask.sum(1).tolist()
        cu = [0]
        for x in lens:
            cu.append(cu[-1] + int(x))
        T = emb.shape[0]
        ids = torch.zeros(T, dtype=torch.long, device=dev)
        slot = torch.arange(T, dtype=torch.int32, device=dev)          # every row comes from `emb`
        return self._greedy(ids, slot, cu, emb, T, max_new, eos, pad, sampler=sampler, processors=procs, beams=beams)

    @torch.no_grad()
    def generate_stage2(self, pixel_values, input_ids, attention_mask=None, image_flags=None, motion_feature=None,
                        generation_config=None, **generate_kwargs) -> torch.LongTensor:
        """Greedy decode behind a stage-2 prompt: the embedding assembly of the reference's ``chat2``
        (modeling_internvl_chat.py:642-707: all <IMG_CONTEXT> slots but the last of each clip take visual tokens, the last one the
        motion token) followed by its ``generate2``.  Ids and slot map go to the native prefill, whose embed kernel gathers
        token / visual / motion rows - no embedding tensor is assembled on the host side."""
        if self.img_context_token_id is None:
            raise AssertionError("img_context_token_id must be set (stage2_eval.py:810)")
        max_new, eos, pad, sampler, procs, beams = self._gen_args(generation_config, generate_kwargs)
        pad = self.config.llm_config.pad_token_id if pad is None else pad
        B = input_ids.shape[0]
        plan = self._plan(input_ids, attention_mask, None, image_flags, pixel_values.shape[0], drop_dead_tail=False)
        motion_feature = self._motion_feature(pixel_values, B, motion_feature)
        self._native(n_frames=pixel_values.shape[0], n_tokens=plan["cu"][-1], n_clips=B)
        vit_embeds, motion = self._visual_inputs(pixel_values, None, motion_feature, plan)
        return self._greedy(plan["ids_packed"], plan["slot"], plan["cu"], vit_embeds, plan["n_vis"], max_new, eos, pad, motion=motion, sampler=sampler, processors=procs, beams=beams)

    def chat2(self, tokenizer, pixel_values, input_ids, generation_config, attention_mask, history=None,
              return_history=False, image_flags=None, IMG_START_TOKEN="<img>", IMG_END_TOKEN="</img>",
              IMG_CONTEXT_TOKEN="<IMG_CONTEXT>", verbose=False, motion_feature=None):
        """modeling_internvl_chat.py:638-767: pre-tokenised stage-2 prompt (with the motion slot) -> decoded response."""
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        template = get_conv_template(self.template)
        generation_config["eos_token_id"] = tokenizer.convert_tokens_to_ids(template.sep)
        out = self.generate_stage2(pixel_values, input_ids, attention_mask, image_flags, motion_feature, **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0].split(template.sep)[0].strip()
        return (response, history) if return_history else response

    def chat(self, tokenizer, pixel_values, question, generation_config, history=None, return_history=False,
             num_patches_list=None, IMG_START_TOKEN="<img>", IMG_END_TOKEN="</img>", IMG_CONTEXT_TOKEN="<IMG_CONTEXT>",
             verbose=False):
        """modeling_internvl_chat.py:582-636 (mutates generation_config['eos_token_id'] like the reference)."""
        if history is None and pixel_values is not None and "<image>" not in question:
            question = "<image>\n" + question
        if num_patches_list is None:
            num_patches_list = [pixel_values.shape[0]] if pixel_values is not None else []
        assert pixel_values is None or len(pixel_values) == sum(num_patches_list)
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        template = get_conv_template(self.template)
        template.system_message = self.system_message
        eos_token_id = tokenizer.convert_tokens_to_ids(template.sep)
        history = [] if history is None else history
        for old_q, old_a in history:
            template.append_message(template.roles[0], old_q)
            template.append_message(template.roles[1], old_a)
        template.append_message(template.roles[0], question)
        template.append_message(template.roles[1], None)
        query = template.get_prompt()
        for num_patches in num_patches_list:
            image_tokens = IMG_START_TOKEN + IMG_CONTEXT_TOKEN * self.num_image_token * num_patches + IMG_END_TOKEN
            query = query.replace("<image>", image_tokens, 1)
        model_inputs = tokenizer(query, return_tensors="pt")
        generation_config["eos_token_id"] = eos_token_id
        out = self.generate(pixel_values=pixel_values, input_ids=model_inputs["input_ids"],
                            attention_mask=model_inputs["attention_mask"], **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0].split(template.sep)[0].strip()
        history.append((question, response))
        if return_history:
            return response, history
        if verbose:
            print(query.replace(IMG_CONTEXT_TOKEN, "").replace(f"{IMG_START_TOKEN}{IMG_END_TOKEN}", "<image>"), response)
        return response

    def batch_chat(self, tokenizer, pixel_values, questions, generation_config, num_patches_list=None, history=None,
                   return_history=False, IMG_START_TOKEN="<img>", IMG_END_TOKEN="</img>",
                   IMG_CONTEXT_TOKEN="<IMG_CONTEXT>", verbose=False, image_counts=None):
        """modeling_internvl_chat.py:533-580 (left padding is stripped by the packed layout)."""
        if history is not None or return_history:
            raise NotImplementedError("Now multi-turn chat is not supported in batch_chat.")
        if image_counts is not None:
            num_patches_list = image_counts
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        queries = []
        template = None
        for idx, num_patches in enumerate(num_patches_list):
            question = questions[idx]
            if pixel_values is not None and "<image>" not in question:
                question = "<image>\n" + question
            template = get_conv_template(self.template)
            template.system_message = self.system_message
            template.append_message(template.roles[0], question)
            template.append_message(template.roles[1], None)
            query = template.get_prompt()
            image_tokens = IMG_START_TOKEN + IMG_CONTEXT_TOKEN * self.num_image_token * num_patches + IMG_END_TOKEN
            queries.append(query.replace("<image>", image_tokens, 1))
        tokenizer.padding_side = "left"
        model_inputs = tokenizer(queries, return_tensors="pt", padding=True)
        generation_config["eos_token_id"] = tokenizer.convert_tokens_to_ids(template.sep)
        out = self.generate(pixel_values=pixel_values, input_ids=model_inputs["input_ids"],
                            attention_mask=model_inputs["attention_mask"], **generation_config)
        responses = tokenizer.batch_decode(out, skip_special_tokens=True)
        return [r.split(template.sep)[0].strip() for r in responses]

    def set_precision(self, mode: str = "bf16"):
        """"bf16" (default: the reference's dtype flow) or "fp8": the InternLM2 prefill linears of ``forward`` on the e4m3 MFMA with
        per-channel weight scales and per-token activation scales (BASELINE config 5; aigv_set_precision in include/aigv_amd.h).
        The reference has no fp8 path; scores move by the quantisation noise documented in DESIGN.md."""
        self._drop_graphs()
        if mode not in ("bf16", "fp8"):
            raise ValueError("precision must be 'bf16' or 'fp8'")
        self._precision = mode
        if self._ctx is not None and not self._dirty:
            native.check(native.load().aigv_set_precision(self._ctx, 1 if mode == "fp8" else 0), self._ctx)

    def set_attention_numerics(self, mode: str = "fp32"):
        """Prefill attention: "fp32" (default since round 5) keeps the score matrix in fp32 up to the softmax; "reference" rounds it to bf16
        exactly where the reference's eager path does (modeling_internlm2.py:417, modeling_intern_vit.py:153).  Over the 37 clips the
        imported reference was recorded on the two are equally far from its bf16 scores (2.80 / 3.09 bf16 ulps mean; the reference against
        itself under other host thread counts: 2.56), "fp32" is closer to its fp32 scores (2.01 / 3.59) and ~1.6 % faster
        (profiles/r5_parity_stats.txt)."""
        self._drop_graphs()
        if mode not in ("reference", "fp32"):
            raise ValueError("attention numerics must be 'reference' or 'fp32'")
        self._attn_numerics = 1 if mode == "reference" else 0
        lib, ctx = self._native()
        native.check(lib.aigv_set_attention_numerics(ctx, self._attn_numerics), ctx)

    def set_gemm_mode(self, mode: int = -1):
        """GEMM tile choice of this model's context (aigv_set_gemm_mode): -1 process default, 0 per-clip / per-frame row plans (the
        default: batch-invariant bits), 1 every row on the 128x128 kernel, 2 the 256x256 kernel wherever it applies (both full K: test
        aliases), 3 the batch-level cost-model dispatch of rounds 1-3 (A/B only)."""
        self._drop_graphs()
        self._gemm_mode = int(mode)
        lib, ctx = self._native()
        native.check(lib.aigv_set_gemm_mode(ctx, int(mode)), ctx)

    TUNE_KNOBS = {"gemm_mode": 0, "gemm256_order": 1, "gemm256_variant": 2, "attn_waves": 3, "skinny_p": 4, "body_tile": 5, "co_kmax": 6,
                  "tail_slices": 7, "attn_lead_key": 8, "decode_fused": 9, "decode_fp8": 10, "skinny_p8": 11, "fuse_tails": 12, "lone_body": 13}

    def tune(self, knob: str, value: int = -1):
        """Experiment knobs of THIS model's context (aigv_ctx_tune; -1 = follow the process default): tests and A/B runs only."""
        self._drop_graphs()
        lib, ctx = self._native()
        native.check(lib.aigv_ctx_tune(ctx, self.TUNE_KNOBS[knob], int(value)), ctx)

    def capture_forward(self, **forward_kwargs):
        """One scoring pass captured into a HIP graph (torch.cuda.CUDAGraph: every launch ``forward`` makes through the C ABI on torch's
        current stream, the SlowFast branch on its side stream, the small index uploads) -> ``(replay, outputs)``: ``replay()`` re-runs the
        ~1000 launches of the pass with ONE host call and refreshes ``outputs`` (the dict ``forward`` returned, static tensors) from the
        CURRENT contents of the input tensors' device memory; host-side arguments (token ids, labels, masks) are frozen at capture time.
        For callers whose host cannot keep up with the launch stream (a CPU-throttled container): same kernels, same bits.  The context
        must be warm (one eager ``forward`` of the same shapes first); profiling brackets must be off.  The replay stays valid until the model's
        weights, modes or capacities change or the motion branch retires the native handle of this geometry (SlowFastR50 keeps MAX_HANDLES
        geometries alive): capture again after any of those (``enable_graph_replay`` tracks all of that by itself)."""
        pv = forward_kwargs.get("pixel_values")
        if torch.is_tensor(pv) and forward_kwargs.get("motion_feature") is None and forward_kwargs.get("input_ids") is not None:
            self._prepare_motion_branch(pv, int(forward_kwargs["input_ids"].shape[0]))
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        self._capture_keep = []
        try:
            with native.capturing(), torch.cuda.graph(graph, capture_error_mode="relaxed"):
                outputs = self.forward(**forward_kwargs)
            keep = self._capture_keep
        finally:
            self._capture_keep = None

        def replay(_graph=graph, _keep=keep):
            _graph.replay()
            return outputs
        return replay, outputs

    def set_row_trimming(self, on: bool = True):
        """Last-layer row trimming (default on): the last decoder layer finishes only the rows whose hidden state is
        consumed (score row + answer rows; stage2_eval.py:940-941, modeling_internvl_chat.py:469-481).  Off = every row
        through every layer, as the reference computes it; the returned values are the same."""
        self._drop_graphs()
        self._row_trim = bool(on)
        lib, ctx = self._native()
        native.check(lib.aigv_set_row_trimming(ctx, int(on)), ctx)
        self.drop_dead_tail = bool(on)      # the host-side half: tokens behind a clip's last consumed row are not run

    # ---- measurement ---------------------------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        self._drop_graphs()
        self._prof_on = bool(on)        # (per-launch HIP events: such passes are not replayed from a graph)
        lib, ctx = self._native()
        native.check(lib.aigv_prof_enable(ctx, int(on)), ctx)

    def prof_read(self) -> Dict[str, Dict[str, float]]:
        lib, ctx = self._native()
        out = {}
        for cls, name in enumerate(("gemm_llm", "attn_vit", "attn_llm", "skinny", "gemm_fp8", "gemm_vit")):
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            native.check(lib.aigv_prof_read(ctx, cls, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), ctx)
            out[name] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        out["gemm"] = {k: out["gemm_llm"][k] + out["gemm_vit"][k] for k in out["gemm_llm"]}     # every bf16 tile-kernel GEMM launch
        return out
