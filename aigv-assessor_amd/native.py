"""ctypes binding of ``libaigv_amd.so`` (C ABI: include/aigv_amd.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIGV_AMD_LIB") or os.path.join(HERE, "libaigv_amd.so")   # (the override: kernel-ablation builds of scripts/)
ABI_VERSION = 1


class NativeError(RuntimeError):
    pass


class AigvConfig(C.Structure):
    """Mirror of ``struct aigv_config`` (include/aigv_amd.h) — field order and types must match."""
    _fields_ = [
        ("vit_hidden", C.c_int32), ("vit_inter", C.c_int32), ("vit_heads", C.c_int32), ("vit_layers", C.c_int32),
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("num_channels", C.c_int32),
        ("vit_norm_rms", C.c_int32), ("vit_qk_norm", C.c_int32), ("vit_qkv_bias", C.c_int32),
        ("vit_eps", C.c_float), ("select_layer", C.c_int32), ("shuffle", C.c_int32),
        ("llm_hidden", C.c_int32), ("llm_inter", C.c_int32), ("llm_heads", C.c_int32), ("llm_kv_heads", C.c_int32),
        ("llm_layers", C.c_int32), ("vocab", C.c_int32), ("rms_eps", C.c_float), ("max_positions", C.c_int32),
        ("motion_dim", C.c_int32), ("n_score_layers", C.c_int32), ("score_dims", C.c_int32 * 8),
        ("max_frames", C.c_int32), ("vit_chunk", C.c_int32), ("max_tokens", C.c_int32), ("max_seqs", C.c_int32),
        ("max_out_rows", C.c_int32), ("kv_capacity", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_I64P = C.POINTER(C.c_int64)
_I32P = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol declared in include/aigv_amd.h
PROTOTYPES = {
    "aigv_abi_version": (_I, []),
    "aigv_sizeof_config": (_I, []),
    "aigv_ctx_create": (_I, [_I, C.POINTER(AigvConfig), C.POINTER(_P)]),
    "aigv_ctx_destroy": (None, [_P]),
    "aigv_ctx_resize": (_I, [_P, C.POINTER(AigvConfig)]),
    "aigv_last_error": (C.c_char_p, [_P]),
    "aigv_clear_hip_error": (None, []),
    "aigv_load_weight": (_I, [_P, C.c_char_p, _P, _I64P, _I, _I, _I]),
    "aigv_finalize_weights": (_I, [_P]),
    "aigv_vit_forward": (_I, [_P, _P, _I, _P, _P]),
    "aigv_project": (_I, [_P, _P, _I, _P, _P]),
    "aigv_motion_project": (_I, [_P, _P, _I, _P, _P]),
    "aigv_llm_prefill": (_I, [_P, _P, _P, _I32P, _I, _P, _I, _P, _I32P, _P, _I32P, _I, _P, _I, _P]),
    "aigv_llm_extend": (_I, [_P, _P, _I32P, _I, _I32P, _P, _I32P, _I, _P, _I, _P]),
    "aigv_kv_fork": (_I, [_P, _I, _P]),
    "aigv_kv_reorder": (_I, [_P, _P, _P, _I, _P]),
    "aigv_set_row_trimming": (_I, [_P, _I]),
    "aigv_set_gemm_mode": (_I, [_P, _I]),
    "aigv_set_attention_numerics": (_I, [_P, _I]),
    "aigv_get_attention_numerics": (_I, [_P]),
    "aigv_ctx_tune": (_I, [_P, _I, _I]),
    "aigv_decode_step": (_I, [_P, _P, _P, _P]),
    "aigv_out_row_logits": (_I, [_P, _I, _I, _P, _I, _P]),
    "aigv_out_row_hidden": (_I, [_P, _I, _I, _P, _I, _P]),
    "aigv_decode_eos": (_I, [_P, _P, _P, _I64P, _I, C.c_int64, _P]),
    "aigv_op_gemm": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "aigv_op_gemm_rows": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, C.POINTER(C.c_int32), _I, _I, _I, _I, _P]),
    "aigv_op_gemm_splitk": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "aigv_op_gemm_splitk256": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "aigv_slowfast_create": (_I, [_I, _I, _I, _I, _I, C.POINTER(_P)]),
    "aigv_slowfast_destroy": (None, [_P]),
    "aigv_slowfast_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I, _I]),
    "aigv_slowfast_finalize": (_I, [_P]),
    "aigv_slowfast_forward": (_I, [_P, _P, _I, _P, _P]),
    "aigv_slowfast_flops_per_clip": (C.c_double, [_P]),
    "aigv_op_conv3d": (_I, [_P, _I, _I, _I, C.POINTER(_I), _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "aigv_set_precision": (_I, [_P, _I]),
    "aigv_op_quant_fp8_rows": (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    "aigv_op_gemm_fp8": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "aigv_op_skinny_gemm": (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P]),
    "aigv_op_skinny_gemm_fp8": (_I, [_P, _I, _I, _P, _I, C.POINTER(C.c_float), _I, _I, _P, _I, _P, _I, _I, _P, _F, _I, _P]),
    "aigv_op_layernorm": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _F, _P]),
    "aigv_op_rmsnorm": (_I, [_P, _I, _P, _P, _I, _I, _I, _F, _P, _P]),
    "aigv_op_rope": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "aigv_op_attention": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "aigv_op_attention_rope": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    "aigv_op_pixel_shuffle": (_I, [_P, _I, _I, _P, _I, _P]),
    "aigv_op_im2col": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "aigv_op_lm_head_argmax": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "aigv_op_frame_ingest": (_I, [_P, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "aigv_op_frame_resize_ingest": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P, _P]),
    "aigv_tune_gemm": (_I, [_I, C.c_double]),
    "aigv_tune_co_gemm": (_I, [_I]),
    "aigv_tune_default": (_I, [_I, _I]),
    "aigv_tune_attention": (_I, [_I]),
    "aigv_tune_skinny": (_I, [_I]),
    "aigv_plan_gemm": (_I, [_I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "aigv_prof_enable": (_I, [_P, _I]),
    "aigv_prof_read": (_I, [_P, _I, _I64P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library and bind every prototype; raises if it is missing or the ABI differs."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
                          "there is no CPU fallback for the product path")
    import torch  # noqa: F401  - before the library: it must bind to the HIP runtime torch has loaded (two runtimes in one process do not share devices)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.aigv_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI mismatch: library {lib.aigv_abi_version()} vs binding {ABI_VERSION}")
    if lib.aigv_sizeof_config() != C.sizeof(AigvConfig):
        raise NativeError(f"aigv_config layout mismatch: C {lib.aigv_sizeof_config()} vs ctypes {C.sizeof(AigvConfig)}")
    _lib = lib
    return lib


def check(rc: int, ctx=None):
    if rc != 0:
        msg = load().aigv_last_error(ctx)
        raise NativeError(f"libaigv_amd error {rc}: {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device (or host) address of a torch tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "native ops take contiguous tensors"
    return t.data_ptr()


def stream_ptr() -> Optional[int]:
    import torch
    s = torch.cuda.current_stream().cuda_stream
    return s if s else None


def i32_array(values):
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr


# ---- releases of native handles and stream captures ------------------------------------------------------------------------------------------------------
# aigv_ctx_destroy / aigv_slowfast_destroy free device memory (hipFree: a device synchronisation).  Inside a stream capture that INVALIDATES the capture, and
# on ROCm 7.2 an invalidated capture cannot be recovered from (scripts/capture_error_probe.py).  A finalizer can run at any moment: the host-side models hold
# reference cycles, so they die in Python's CYCLIC collector - which torch 2.10 no longer runs in front of a capture (torch.cuda.graph.__enter__,
# force_cudagraph_gc) and which may therefore fire in the middle of one (seen as a now-and-then failure of whichever capture came after a model had gone out of
# scope).  So: every capture this package starts runs inside `capturing()` - cyclic garbage collected first, the collector off for the duration - and a
# release requested while a capture is underway is parked and carried out afterwards.
import contextlib
import gc

_captures_underway = 0
_deferred_releases = []


def release(fn_name: str, handle) -> None:
    """Destroy a native handle now, or - while a stream capture is underway - after it."""
    if handle is None:
        return
    capturing = _captures_underway > 0
    if not capturing:
        try:
            import torch
            capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        except Exception:
            capturing = False
    if capturing:
        _deferred_releases.append((fn_name, handle))
    else:
        getattr(load(), fn_name)(handle)


def flush_releases() -> None:
    while _deferred_releases and _captures_underway == 0:
        fn_name, handle = _deferred_releases.pop()
        getattr(load(), fn_name)(handle)


@contextlib.contextmanager
def capturing():
    """Bracket of every stream capture this package starts (InternVLChatModel._graph_call / capture_forward)."""
    global _captures_underway
    gc.collect()                      # whatever is garbage now dies now, outside the capture
    was_enabled = gc.isenabled()
    gc.disable()
    _captures_underway += 1
    try:
        yield
    finally:
        _captures_underway -= 1
        if was_enabled:
            gc.enable()
        flush_releases()
