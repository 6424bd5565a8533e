"""Rotary position embedding on packed-head tensors ``[B, T, H, hs]`` (interleaved pairs).

Tables are real ``cos``/``sin`` fp32 ``[block, hs/2]`` (the reference keeps a complex64
``freqs_cis``; the checkpoint writer regenerates that buffer so the key set stays identical).
Rotation math is done in fp32 and cast back, as in the reference
(reference: single-gpu/model.py:77-96,566-577).
"""
from __future__ import annotations

import torch

from .._ext import ops, use_hip


def rope_tables(block_size: int, dim: int, base: float = 10000.0, device=None):
    theta = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    freqs = torch.outer(torch.arange(block_size, dtype=torch.float32), theta)
    return torch.cos(freqs).to(device), torch.sin(freqs).to(device)


def _rope_eager(x, cos_t, sin_t, pos0: int, inverse: bool):
    B, T, H, hs = x.shape
    xf = x.float().reshape(B, T, H, hs // 2, 2)
    re, im = xf.unbind(-1)
    c = cos_t[pos0:pos0 + T].view(1, T, 1, hs // 2)
    s = sin_t[pos0:pos0 + T].view(1, T, 1, hs // 2)
    if inverse:
        s = -s
    out = torch.stack([re * c - im * s, re * s + im * c], dim=-1).flatten(3)
    return out.type_as(x)


class _RopeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cos_t, sin_t, pos0):
        y = x.contiguous().clone() if x.is_contiguous() else x.contiguous()
        ops().rope_(y, cos_t, sin_t, pos0, False)
        ctx.save_for_backward(cos_t, sin_t)
        ctx.pos0 = pos0
        return y

    @staticmethod
    def backward(ctx, dy):
        cos_t, sin_t = ctx.saved_tensors
        dx = dy.contiguous().clone() if dy.is_contiguous() else dy.contiguous()
        ops().rope_(dx, cos_t, sin_t, ctx.pos0, True)
        return dx, None, None, None


def apply_rope(x: torch.Tensor, cos_t: torch.Tensor, sin_t: torch.Tensor, pos0: int = 0) -> torch.Tensor:
    """Rotate ``x [B, T, H, hs]`` by the angles of positions ``pos0 .. pos0+T-1``."""
    hs = x.shape[-1]
    if use_hip(x, cos_t, sin_t) and hs % 8 == 0 and x.dtype in (torch.bfloat16, torch.float32):
        return _RopeFn.apply(x, cos_t, sin_t, int(pos0))
    return _rope_eager(x, cos_t, sin_t, int(pos0), False)


def rope_inplace_(x: torch.Tensor, cos_t, sin_t, pos0: int = 0, inverse: bool = False) -> torch.Tensor:
    """In-place rotation of a (possibly strided) ``[B, T, H, hs]`` view; no autograd."""
    if use_hip(x, cos_t, sin_t) and x.shape[-1] % 8 == 0:
        ops().rope_(x, cos_t, sin_t, int(pos0), bool(inverse))
    else:
        x.copy_(_rope_eager(x, cos_t, sin_t, int(pos0), inverse))
    return x
