"""CPU ORACLE for the build's fp8 mode of the InternLM2 prefill linears  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference has NO fp8 path: this file restates the arithmetic `aigv_set_precision(ctx, AIGV_PRECISION_FP8_LLM)` defines
(include/aigv_amd.h; BASELINE.json config 5, "InternVL2-8B fp8 weights (CDNA4 fp8 MFMA)"), so that the HIP kernels can be checked
against an independent evaluation of the same definition.  It says nothing about closeness to the reference - that is measured
separately as the drift between the fp8 and the bf16 results (tests/test_gpu_e2e.py).

Definition (per linear y = x W^T with bf16 x [rows, K] and W [N, K]):
    scale_r = amax_k |x[r, k]| / 448 (1 for an all-zero row);  qx[r, k] = e4m3_rne(x[r, k] * (448 / amax_r))      (OCP e4m3fn)
    the same per output channel n of W;  acc[r, n] = sum_k qx[r, k] qw[n, k] in fp32 (products exact);
    y[r, n] = bf16((acc * scale_r) * scale_n)
applied to wqkv, wo, w1, w3, w2 of every decoder layer except wo / w1 / w3 / w2 of the LAST layer (they act on the few consumed
rows and stay bf16).  Everything else is oracle.py's bf16 flow.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

from . import oracle as O

Tensor = torch.Tensor


def quant_rows(x: Tensor):
    """bf16/fp32 [rows, K] -> (float8_e4m3fn [rows, K], fp32 scale [rows]) with three fp32 operations per element."""
    x = x.float()
    amax = x.abs().amax(dim=-1, keepdim=True)
    inv = torch.where(amax > 0, torch.full_like(amax, 448.0) / amax, torch.ones_like(amax))     # a true IEEE division
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (x * inv).to(torch.float8_e4m3fn), scale.reshape(-1)


_WEIGHT_CACHE = None          # inside fp8_llm(): {id of the weight's storage: (dequantised e4m3 weight as fp32, channel scales)}
_WEIGHT_CACHE_CAP = 6 << 30   # bytes of cached fp32 copies (a 3-layer 8B-width test: 2.6 GB; a full-depth run exceeds it and recomputes)


def _quant_weight(w: Tensor):
    """The weight side of fp8_linear.  The quantised copy of a weight is a function of the weight alone, so inside one fp8_llm() context it
    is computed once per tensor (a KV-cache decode loop calls every linear once per token: re-quantising 218 M parameters per layer and
    token was most of tests/test_gpu_e2e.py::test_decode_at_8b_widths's wall time)."""
    global _WEIGHT_CACHE
    key = (w.data_ptr(), tuple(w.shape))
    if _WEIGHT_CACHE is not None and key in _WEIGHT_CACHE:
        return _WEIGHT_CACHE[key][:2]
    qw, sw = quant_rows(w)
    qf = qw.float()
    if _WEIGHT_CACHE is not None and sum(v[2] for v in _WEIGHT_CACHE.values()) + qf.numel() * 4 <= _WEIGHT_CACHE_CAP:
        _WEIGHT_CACHE[key] = (qf, sw, qf.numel() * 4, w)       # (holds `w` so that the address cannot be recycled while cached)
    return qf, sw


def fp8_linear(x: Tensor, w: Tensor) -> Tensor:
    shp = x.shape
    qa, sa = quant_rows(x.reshape(-1, shp[-1]))
    qf, sw = _quant_weight(w)
    acc = qa.float() @ qf.t()
    return ((acc * sa[:, None]) * sw[None, :]).to(x.dtype).reshape(*shp[:-1], w.shape[0])


@contextlib.contextmanager
def fp8_llm(n_layers: int):
    """Inside this context oracle.llm_forward evaluates the linears of the fp8 mode in fp8."""
    def hook(x, w, layer, name):
        if layer == n_layers - 1 and name != "wqkv":
            return F.linear(x, w)
        return fp8_linear(x, w)

    global _WEIGHT_CACHE
    prev, prev_cache = O.LLM_LINEAR_HOOK, _WEIGHT_CACHE
    O.LLM_LINEAR_HOOK = hook
    _WEIGHT_CACHE = {}
    try:
        yield
    finally:
        O.LLM_LINEAR_HOOK = prev
        _WEIGHT_CACHE = prev_cache
