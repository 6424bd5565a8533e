"""Causal attention over packed-head tensors, GQA-native, with bottom-right aligned masking.

``causal_attention(q, k, v, scale)`` takes ``q [B, Tq, nh, hs]`` and ``k, v [B, Tk, nkv, hs]``
(any batch/token strides, heads packed — e.g. slices of a fused QKV projection) and returns
``o [B, Tq, nh, hs]``. Query ``i`` attends to keys ``<= i + Tk - Tq``; with ``Tq == Tk`` this is
the usual causal mask and with a KV cache (``Tq < Tk``) the new tokens see the whole cache.

GPU path: the hand-written flash kernels (bf16, hs in {64, 128}).
Eager path (CPU, fp32 checks, other head sizes): explicit-mask SDPA.

Reference behaviour: ``F.scaled_dot_product_attention(q, k, v, is_causal=True)`` after
``repeat_interleave`` of the KV heads (reference: single-gpu/model.py:144-149). The reference's
cached decode is top-left aligned (a 1-token query sees only key 0); that is a bug we do not
reproduce.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip


def _hip_ok(q, k, v) -> bool:
    return (use_hip(q, k, v) and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16
            and v.dtype == torch.bfloat16 and q.shape[-1] in (64, 128) and k.shape[-1] == q.shape[-1]
            and v.shape[-1] == q.shape[-1] and q.shape[1] >= 1)


def _paddable(q, k, v) -> bool:
    """bf16 GPU tensors whose head sizes are not native to the kernels but fit after zero padding."""
    return (use_hip(q, k, v) and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16
            and q.shape[-1] == k.shape[-1] and max(q.shape[-1], v.shape[-1]) <= 128 and q.shape[1] >= 1
            and k.shape[2] == v.shape[2] and q.shape[2] % k.shape[2] == 0)


def _packed(t: torch.Tensor) -> torch.Tensor:
    """Make heads packed (stride(3)==1, stride(2)==hs) and rows 16-byte aligned."""
    hs = t.shape[-1]
    ok = (t.stride(3) == 1 and (t.shape[2] == 1 or t.stride(2) == hs) and t.stride(1) % 8 == 0
          and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0)
    return t if ok else t.contiguous()


class _FlashAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        q, k, v = _packed(q), _packed(k), _packed(v)
        o, lse = ops().attn_fwd(q, k, v, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, Tq, nh, hs = q.shape
        Tk, nkv = k.shape[1], k.shape[2]
        do = _packed(do)
        if Tq == Tk:
            # One fused [B, T, (nh + 2 nkv) hs] gradient buffer: the QKV projection's backward
            # GEMMs consume it without a concatenation.
            dqkv = torch.empty(B, Tq, (nh + 2 * nkv) * hs, dtype=q.dtype, device=q.device)
            dq = dqkv[..., : nh * hs].view(B, Tq, nh, hs)
            dk = dqkv[..., nh * hs: (nh + nkv) * hs].view(B, Tk, nkv, hs)
            dv = dqkv[..., (nh + nkv) * hs:].view(B, Tk, nkv, hs)
        else:
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops().attn_bwd(do, q, k, v, o, lse, dq, dk, dv, ctx.scale)
        return dq, dk, dv, None


def split_qkv(qkv: torch.Tensor, nh: int, nkv: int):
    """Views ``q [B,T,nh,hs]``, ``k, v [B,T,nkv,hs]`` of a fused projection ``[B,T,(nh+2nkv)hs]``."""
    B, T, W = qkv.shape
    hs = W // (nh + 2 * nkv)
    q = qkv[..., : nh * hs].view(B, T, nh, hs)
    k = qkv[..., nh * hs: (nh + nkv) * hs].view(B, T, nkv, hs)
    v = qkv[..., (nh + nkv) * hs:].view(B, T, nkv, hs)
    return q, k, v


class _FlashAttnQKVFn(torch.autograd.Function):
    """Self-attention straight from the fused QKV projection; the backward writes dq, dk, dv into
    one fused buffer that is returned as the gradient of ``qkv`` (no split/cat copies)."""

    @staticmethod
    def forward(ctx, qkv, nh, nkv, scale):
        qkv = qkv if (qkv.is_contiguous() and qkv.data_ptr() % 16 == 0) else qkv.contiguous()
        q, k, v = split_qkv(qkv, nh, nkv)
        o, lse = ops().attn_fwd(q, k, v, scale)
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (nh, nkv, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        nh, nkv, scale = ctx.cfg
        q, k, v = split_qkv(qkv, nh, nkv)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = split_qkv(dqkv, nh, nkv)
        ops().attn_bwd(_packed(do), q, k, v, o, lse, dq, dk, dv, scale)
        return dqkv, None, None, None


def causal_self_attention_qkv(qkv: torch.Tensor, nh: int, nkv: int, scale: float | None = None,
                              dropout_p: float = 0.0) -> torch.Tensor:
    """``qkv [B, T, (nh + 2 nkv) hs]`` -> ``o [B, T, nh, hs]`` (training path, no cache)."""
    hs = qkv.shape[-1] // (nh + 2 * nkv)
    if scale is None:
        scale = 1.0 / math.sqrt(hs)
    if dropout_p == 0.0 and use_hip(qkv) and qkv.dtype == torch.bfloat16 and hs in (64, 128):
        return _FlashAttnQKVFn.apply(qkv, nh, nkv, float(scale))
    q, k, v = split_qkv(qkv, nh, nkv)
    return causal_attention_eager(q, k, v, float(scale), dropout_p)


def causal_attention_eager(q, k, v, scale: float, dropout_p: float = 0.0) -> torch.Tensor:
    B, Tq, nh, hs = q.shape
    Tk, nkv = k.shape[1], k.shape[2]
    qh, kh, vh = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    if nkv != nh:
        kh = kh.repeat_interleave(nh // nkv, dim=1)
        vh = vh.repeat_interleave(nh // nkv, dim=1)
    if Tq == Tk:
        y = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=dropout_p, is_causal=True, scale=scale)
    else:
        mask = torch.ones(Tq, Tk, dtype=torch.bool, device=q.device).tril(diagonal=Tk - Tq)
        y = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask, dropout_p=dropout_p, scale=scale)
    return y.transpose(1, 2)


def causal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float | None = None,
                     dropout_p: float = 0.0) -> torch.Tensor:
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if dropout_p == 0.0 and _hip_ok(q, k, v):
        return _FlashAttnFn.apply(q, k, v, float(scale))
    if dropout_p == 0.0 and _paddable(q, k, v):
        # Other head sizes up to 128 run on the flash kernels too: zero features change neither the
        # scores (the scale is explicit) nor the kept output columns, and no T x T matrix is formed.
        hs, dv = q.shape[-1], v.shape[-1]
        tgt = 64 if max(hs, dv) <= 64 else 128
        qp, kp = F.pad(q, (0, tgt - hs)), F.pad(k, (0, tgt - hs))
        vp = F.pad(v, (0, tgt - dv))
        return _FlashAttnFn.apply(qp, kp, vp, float(scale))[..., :dv]
    return causal_attention_eager(q, k, v, float(scale), dropout_p)
