"""Decode-step linear layer: ``y = act(LN(x) @ w.T + bias) + residual`` for at most 64 rows in ONE launch
(``csrc/decode.hip``), optionally writing the key / value columns of a fused qkv projection straight into
the preallocated KV cache at the device-side position.

A cached decode step of a dense block is then five launches (qkv, attention, output projection, c_fc,
c_proj) instead of about ten (LayerNorm, library GEMM, bias, activation and residual add each on their
own), which is what bounds the per-token latency of a small model.

Reference behaviour: the ``nn.Linear`` / ``nn.LayerNorm`` / activation calls a block makes for one token
(reference: single-gpu/model.py:508-533, generate loop :699-747).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip

ACT_NONE, ACT_GELU, ACT_SWIGLU = 0, 1, 2


def decode_linear_supported(x2: torch.Tensor, w: torch.Tensor, act: int = ACT_NONE, has_ln: bool = False) -> bool:
    """``x2 [B <= 64, K]`` and ``w [N (2N for SwiGLU), K]`` bf16 on the GPU, K % 128 == 0 (and at most 2048
    with the LayerNorm prologue), N % 16 == 0."""
    if os.environ.get("DPA_FUSED_DECODE", "1") == "0":
        return False
    if not (use_hip(x2, w) and x2.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x2.dim() == 2):
        return False
    return hasattr(ops(), "decode_linear") and bool(ops().decode_linear_supported(x2, w, int(act), bool(has_ln)))


def decode_linear_eager(x2, ln, w, bias, residual, act):
    """The same computation with stock ops (fp32 checks, CPU)."""
    h = x2
    if ln is not None:
        h = F.layer_norm(h, (h.shape[-1],), ln[0], ln[1], ln[2])
    y = F.linear(h, w, bias)
    if act == ACT_GELU:
        y = F.gelu(y)
    elif act == ACT_SWIGLU:
        g, u = y.chunk(2, dim=-1)
        y = F.silu(g) * u
    return y if residual is None else y + residual


def decode_linear(x2: torch.Tensor, ln, w: torch.Tensor, bias=None, residual=None, act: int = ACT_NONE,
                  kv=None, rope=None) -> torch.Tensor:
    """``ln``: ``(weight, bias | None, eps)`` or None. ``kv``: ``(k_cache, v_cache, pos, kv_col0)``: the
    output columns from ``kv_col0`` on are appended to the caches at row ``pos`` (int64 ``[1]`` on the
    device) and only the columns below ``kv_col0`` are returned. ``rope``: ``(cos_row, sin_row, rope_cols,
    hs)``: the heads (``hs`` columns each) below column ``rope_cols`` are rotated by the fp32 ``[hs / 2]``
    angle rows of the token's position before anything is stored. No gradient."""
    lw, lb, eps = (ln[0], ln[1], float(ln[2])) if ln is not None else (None, None, 0.0)
    k_cache, v_cache, pos, col0 = kv if kv is not None else (None, None, None, 0)
    rc, rs, rcols, hs = rope if rope is not None else (None, None, 0, 0)
    return ops().decode_linear(x2, lw, lb, eps, w, bias, residual, int(act), k_cache, v_cache, pos, int(col0),
                               rc, rs, int(rcols), int(hs))
