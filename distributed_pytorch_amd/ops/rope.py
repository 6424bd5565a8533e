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


class _RopeQkvInplaceFn(torch.autograd.Function):
    """Rotate the leading ``heads`` heads of a fused projection ``[B, T, (heads + rest) * hs]`` IN PLACE
    (the q and k heads of a qkv tensor the caller owns). The gradient is rotated back on a copy: autograd
    does not promise that an incoming gradient belongs to this node alone (retain_graph and a second backward,
    tensor hooks, a producer that keeps its buffer), and this path serves the attention fallbacks whose backward
    arrives through split / cat nodes. The training path that does own its gradient buffer rotates inside the
    attention backward kernels instead (``ops/attention.py::qkv_rope_attention``)."""

    @staticmethod
    def forward(ctx, qkv, cos_t, sin_t, pos0, heads, hs):
        B, T, _ = qkv.shape
        ops().rope_(qkv[..., : heads * hs].view(B, T, heads, hs), cos_t, sin_t, pos0, False)
        ctx.mark_dirty(qkv)
        ctx.save_for_backward(cos_t, sin_t)
        ctx.meta = (pos0, heads, hs)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        cos_t, sin_t = ctx.saved_tensors
        pos0, heads, hs = ctx.meta
        d = dqkv.clone(memory_format=torch.contiguous_format)
        B, T, _ = d.shape
        ops().rope_(d[..., : heads * hs].view(B, T, heads, hs), cos_t, sin_t, pos0, True)
        return d, None, None, None, None, None


def rope_qkv_(qkv: torch.Tensor, cos_t: torch.Tensor, sin_t: torch.Tensor, pos0: int, heads: int, hs: int) -> torch.Tensor:
    """Rotary embedding of the first ``heads`` heads of a fused ``[B, T, W]`` projection, without the copy
    of the rotated slice and the re-concatenation: one in-place pass over the q and k columns in
    forward, one over a copy of their gradient in backward. ``qkv`` must be owned by the caller (a fresh GEMM
    output that nothing else reads)."""
    ok = (use_hip(qkv, cos_t, sin_t) and qkv.dim() == 3 and qkv.is_contiguous() and hs % 8 == 0
          and qkv.dtype in (torch.bfloat16, torch.float32) and (qkv.shape[-1] * qkv.element_size()) % 16 == 0)
    if ok and (not qkv.requires_grad or not qkv.is_leaf):
        return _RopeQkvInplaceFn.apply(qkv, cos_t, sin_t, int(pos0), int(heads), int(hs))
    B, T, _ = qkv.shape
    qk = apply_rope(qkv[..., : heads * hs].reshape(B, T, heads, hs), cos_t, sin_t, pos0)
    return torch.cat([qk.view(B, T, heads * hs), qkv[..., heads * hs:]], dim=-1)
