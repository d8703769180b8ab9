"""Host side of the SlowFast-R50 motion branch.

Mirror of the reference's ``slowfast`` module + ``pack_pathway_output``
(internvl/model/internvl_chat_eval2/modeling_internvl_chat.py:97-193): it is callable with the reference's
``[slow_pathway, fast_pathway]`` list and returns ``[B, 2304, 1, 1, 1]``, and - the form the scorer uses - takes the
``pixel_values`` tensor directly (``features``), because the native branch samples the slow pathway itself.  All compute
is the HIP library (``aigv_slowfast_*`` in include/aigv_amd.h); there is no torch fallback.

The weights are pytorchvideo's ``slowfast_r50`` blocks 0..4 under the reference's state-dict path
``slowfast_model.feature_extraction.``; the reference downloads them at construction time (:164), here they come
from the checkpoint's state dict or a user-supplied file.
"""
from __future__ import annotations

import collections
import ctypes as C
import itertools
from typing import Dict, Iterator, Optional, Tuple

import torch

from . import native

PREFIX = "slowfast_model.feature_extraction."
FEATURE_DIM = 2304


_UIDS = itertools.count(1)


class SlowFastR50:
    MAX_HANDLES = 8      # native handles kept alive, one per (device, T, H, W) geometry, least recently used first out (~0.4 GB each at 4 clips x 8 x 448 x 448)

    def __init__(self, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        self._sd: Dict[str, torch.Tensor] = {}
        # A native handle owns the activation buffers of ONE geometry at a clip capacity.  A model that replays captured HIP graphs has the
        # addresses of those buffers inside its graphs, so a handle must not die under them: handles are cached per geometry (a loop that
        # alternates between frame counts keeps both alive), and ``epoch`` counts every destruction - InternVLChatModel compares it before
        # it replays anything and drops its graphs when it moved (found by tests/manual/fuzz_batched.py: a replay through a destroyed
        # handle's buffers was a GPU memory fault).
        self._handles: "collections.OrderedDict[Tuple, Tuple[int, int]]" = collections.OrderedDict()      # geometry -> (handle, clip capacity)
        self._handle: Optional[int] = None      # the handle of the last call
        self.epoch = 0
        self.uid = next(_UIDS)                  # what graph caches key on (id() of a dead object can come back with the next one)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- weights -------------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Accepts names under ``slowfast_model.feature_extraction.``, ``feature_extraction.`` or pytorchvideo's ``blocks.``."""
        sd = {}
        for k, v in state_dict.items():
            for pre in (PREFIX, "feature_extraction.", "blocks."):
                if k.startswith(pre):
                    k = k[len(pre):]
                    break
            if k.endswith("num_batches_tracked") or not k[:1].isdigit():
                continue
            if int(k.split(".", 1)[0]) > 4:          # blocks 5/6 (pools, classifier) carry nothing the reference keeps
                continue
            sd[k] = v.detach().to(device="cpu", dtype=torch.float32).contiguous()
        if strict and not sd:
            raise RuntimeError("no SlowFast tensors in the state dict")
        self._sd = sd
        self._release()

    def state_dict(self, prefix: str = PREFIX) -> Dict[str, torch.Tensor]:
        return {prefix + k: v for k, v in self._sd.items()}

    def parameters(self) -> Iterator[torch.nn.Parameter]:
        """For the eval driver's freeze loop (stage2_eval.py:857-885): frozen views of the host copies."""
        for v in self._sd.values():
            yield torch.nn.Parameter(v, requires_grad=False)

    def eval(self):
        return self

    # ---- native handle ---------------------------------------------------------------------------------------------------
    def _destroy(self, handle):
        self.epoch += 1
        native.release("aigv_slowfast_destroy", handle)      # (parked while a stream capture is underway: native.release)

    def _release(self):
        """Destroy every native handle (a weight reload, the end of the object)."""
        for h, _cap in list(self._handles.values()):
            self._destroy(h)
        self._handles.clear()
        self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def prepare(self, device: torch.device, clips: int, T: int, H: int, W: int):
        """The native handle for ``clips`` clips of T frames of H x W on ``device``, created (weights uploaded) if there is none yet or the
        cached one is too small.  Creating one allocates and copies synchronously: it cannot happen inside a stream capture, so a caller
        that captures calls this first (InternVLChatModel does) - and compares ``epoch`` afterwards."""
        lib = native.load()
        dev = device.index if device.index is not None else torch.cuda.current_device()
        key = (dev, T, H, W)
        ent = self._handles.get(key)
        if ent is not None and ent[1] >= clips:
            self._handles.move_to_end(key)
            self._handle = ent[0]
            return lib, ent[0]
        if native._captures_underway > 0 or torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"SlowFastR50: no native handle for {clips} clips of {T} x {H} x {W} yet and one cannot be created inside a stream capture "
                               "(call prepare() before capturing)")
        if not self._sd:
            raise RuntimeError("SlowFastR50 has no weights: load_state_dict() first")
        if ent is not None:                                  # the same geometry at a larger clip count: the old handle goes
            del self._handles[key]
            self._destroy(ent[0])
        while len(self._handles) >= self.MAX_HANDLES:
            _k, (h_old, _cap) = self._handles.popitem(last=False)
            self._destroy(h_old)
        h = C.c_void_p()
        native.check(lib.aigv_slowfast_create(dev, clips, T, H, W, C.byref(h)))
        try:
            for name, t in self._sd.items():
                shape = (C.c_int64 * t.dim())(*t.shape)
                native.check(lib.aigv_slowfast_load_weight(h, name.encode(), t.data_ptr(), shape, t.dim(), 1))
            native.check(lib.aigv_slowfast_finalize(h))
        except Exception:
            lib.aigv_slowfast_destroy(h)
            raise
        self._handles[key] = (h, clips)
        self._handle = h
        return lib, h

    def _native(self, device: torch.device, clips: int, T: int, H: int, W: int):
        return self.prepare(device, clips, T, H, W)

    # ---- forward -----------------------------------------------------------------------------------------------------------
    def features(self, pixel_values: torch.Tensor, clips: int) -> torch.Tensor:
        """pixel_values [clips * T, 3, H, W] bf16 on the GPU (clip-major) -> motion feature [clips, 2304] bf16."""
        if not pixel_values.is_cuda:
            raise RuntimeError("SlowFastR50 runs on the GPU only (no CPU fallback)")
        if pixel_values.dim() != 4 or pixel_values.shape[1] != 3 or clips <= 0 or pixel_values.shape[0] % clips:
            raise ValueError(f"pixel_values must be [clips * T, 3, H, W]; got {tuple(pixel_values.shape)} for {clips} clips")
        T, H, W = pixel_values.shape[0] // clips, pixel_values.shape[2], pixel_values.shape[3]
        x = pixel_values.to(torch.bfloat16).contiguous()
        lib, h = self._native(x.device, clips, T, H, W)
        out = torch.empty((clips, FEATURE_DIM), dtype=torch.bfloat16, device=x.device)
        native.check(lib.aigv_slowfast_forward(h, x.data_ptr(), clips, out.data_ptr(), native.stream_ptr()))
        return out

    def flops_per_clip(self) -> float:
        return float(native.load().aigv_slowfast_flops_per_clip(self._handle)) if self._handle is not None else 0.0

    def __call__(self, inputs) -> torch.Tensor:
        """The reference's call form (:179-193): ``[slow, fast]`` with fast = [B, 3, T, H, W]; the slow tensor is ignored because it is,
        by construction (:109-115), ``fast.index_select(2, linspace(0, T-1, T//4).long())``, which the native branch samples itself."""
        fast = inputs[1]
        B, _, T, H, W = fast.shape
        frames = fast.permute(0, 2, 1, 3, 4).reshape(B * T, 3, H, W)
        return self.features(frames, B).view(B, FEATURE_DIM, 1, 1, 1)
