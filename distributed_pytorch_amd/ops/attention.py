"""Causal attention over packed-head tensors, GQA-native, with bottom-right aligned masking.

``causal_attention(q, k, v, scale)`` takes ``q [B, Tq, nh, hs]`` and ``k, v [B, Tk, nkv, hs]``
(any batch/token strides, heads packed — e.g. slices of a fused QKV projection) and returns
``o [B, Tq, nh, hs]``. Query ``i`` attends to keys ``<= i + Tk - Tq``; with ``Tq == Tk`` this is
the usual causal mask and with a KV cache (``Tq < Tk``) the new tokens see the whole cache.

GPU path: the hand-written flash kernels (bf16; key / value widths (64, 64), (128, 128), (128, 64) and
(256, 128) — the last two for latent attention with decoupled rotary keys; other widths zero-padded up).
Attention dropout runs inside them: the keep mask is a stateless hash of (seed, head, query, key)
that the forward and both backward kernels regenerate, the seed is one device integer drawn from
torch's GPU generator (so ``manual_seed``, activation recomputation and hipGraph replay treat it like
any other random op). Eager path (CPU, fp32 checks): explicit-mask SDPA.

Reference behaviour: ``F.scaled_dot_product_attention(q, k, v, is_causal=True)`` after
``repeat_interleave`` of the KV heads (reference: single-gpu/model.py:144-149). The reference's
cached decode is top-left aligned (a 1-token query sees only key 0); that is a bug we do not
reproduce.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip


_WIDTHS = ((64, 64), (128, 128), (128, 64), (256, 128))   # (key width, value width) the kernels are built for


def _hip_ok(q, k, v) -> bool:
    return (use_hip(q, k, v) and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16
            and v.dtype == torch.bfloat16 and (q.shape[-1], v.shape[-1]) in _WIDTHS and k.shape[-1] == q.shape[-1]
            and q.shape[1] >= 1)


def _paddable(q, k, v) -> bool:
    """bf16 GPU tensors whose head sizes are not native to the kernels but fit after zero padding."""
    return (use_hip(q, k, v) and q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16 and v.dtype == torch.bfloat16
            and q.shape[-1] == k.shape[-1] and any(a >= q.shape[-1] and b >= v.shape[-1] for a, b in _WIDTHS)
            and q.shape[1] >= 1 and k.shape[2] == v.shape[2] and q.shape[2] % k.shape[2] == 0)


def _packed(t: torch.Tensor) -> torch.Tensor:
    """Make heads packed (stride(3)==1, stride(2)==hs) and rows 16-byte aligned."""
    hs = t.shape[-1]
    ok = (t.stride(3) == 1 and (t.shape[2] == 1 or t.stride(2) == hs) and t.stride(1) % 8 == 0
          and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0)
    return t if ok else t.contiguous()


def _draw_seed(dropout_p: float, device) -> torch.Tensor | None:
    """One int64 on the device from torch's generator of that device (no host round trip)."""
    if dropout_p <= 0.0:
        return None
    return torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64, device=device)


def dropout_keep_mask(B: int, nh: int, Tq: int, Tk: int, dropout_p: float, seed: torch.Tensor) -> torch.Tensor:
    """The boolean keep mask ``[B, nh, Tq, Tk]`` the kernels apply for ``seed`` (tests, debugging)."""
    return ops().attn_dropout_mask(B, nh, Tq, Tk, float(dropout_p), seed)


class _FlashAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale, dropout_p=0.0, seed=None):
        q, k, v = _packed(q), _packed(k), _packed(v)
        if seed is None:
            seed = _draw_seed(dropout_p, q.device)
        o, lse = ops().attn_fwd(q, k, v, scale, dropout_p, seed)
        ctx.save_for_backward(q, k, v, o, lse, seed)
        ctx.scale = scale
        ctx.dropout_p = dropout_p
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, seed = ctx.saved_tensors
        B, Tq, nh, hs = q.shape
        Tk, nkv = k.shape[1], k.shape[2]
        do = _packed(do)
        if Tq == Tk and v.shape[-1] == hs:
            # One fused [B, T, (nh + 2 nkv) hs] gradient buffer: the QKV projection's backward
            # GEMMs consume it without a concatenation.
            dqkv = torch.empty(B, Tq, (nh + 2 * nkv) * hs, dtype=q.dtype, device=q.device)
            dq = dqkv[..., : nh * hs].view(B, Tq, nh, hs)
            dk = dqkv[..., nh * hs: (nh + nkv) * hs].view(B, Tk, nkv, hs)
            dv = dqkv[..., (nh + nkv) * hs:].view(B, Tk, nkv, hs)
        else:
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops().attn_bwd(do, q, k, v, o, lse, dq, dk, dv, ctx.scale, ctx.dropout_p, seed)
        return dq, dk, dv, None, None, None


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
    def forward(ctx, qkv, nh, nkv, scale, dropout_p=0.0):
        qkv = qkv if (qkv.is_contiguous() and qkv.data_ptr() % 16 == 0) else qkv.contiguous()
        q, k, v = split_qkv(qkv, nh, nkv)
        seed = _draw_seed(dropout_p, qkv.device)
        o, lse = ops().attn_fwd(q, k, v, scale, dropout_p, seed)
        ctx.save_for_backward(qkv, o, lse, seed)
        ctx.cfg = (nh, nkv, scale, dropout_p)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, seed = ctx.saved_tensors
        nh, nkv, scale, dropout_p = ctx.cfg
        q, k, v = split_qkv(qkv, nh, nkv)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = split_qkv(dqkv, nh, nkv)
        ops().attn_bwd(_packed(do), q, k, v, o, lse, dq, dk, dv, scale, dropout_p, seed)
        return dqkv, None, None, None, None


def causal_self_attention_qkv(qkv: torch.Tensor, nh: int, nkv: int, scale: float | None = None,
                              dropout_p: float = 0.0) -> torch.Tensor:
    """``qkv [B, T, (nh + 2 nkv) hs]`` -> ``o [B, T, nh, hs]`` (training path, no cache)."""
    hs = qkv.shape[-1] // (nh + 2 * nkv)
    if scale is None:
        scale = 1.0 / math.sqrt(hs)
    if use_hip(qkv) and qkv.dtype == torch.bfloat16 and hs in (64, 128):
        return _FlashAttnQKVFn.apply(qkv, nh, nkv, float(scale), float(dropout_p))
    q, k, v = split_qkv(qkv, nh, nkv)
    return causal_attention_eager(q, k, v, float(scale), dropout_p)


class _QkvRopeAttnFn(torch.autograd.Function):
    """Self-attention block front end with rotary embedding, without a rotation pass of its own:
    ``o = attention(rope(x W^T + b))``. The projection GEMM rotates the q and k columns in its epilogue
    (``gemm_nt_rope``), and the attention backward kernels rotate dq and dk back as they store them, so the
    fused gradient buffer that reaches the projection's backward GEMMs is already the gradient of the
    unrotated projection (reference: single-gpu/model.py:77-96,130-133)."""

    @staticmethod
    def forward(ctx, x, w, b, cos_t, sin_t, pos0, nh, nkv, hs, scale, dropout_p, sole_use):
        x2 = x.reshape(-1, x.shape[-1])
        B, T = x.shape[0], x.shape[1]
        qkv = ops().gemm_nt_rope(x2, w, b, cos_t, sin_t, pos0, T, (nh + nkv) * hs, hs).view(B, T, -1)
        q, k, v = split_qkv(qkv, nh, nkv)
        seed = _draw_seed(dropout_p, x.device)
        o, lse = ops().attn_fwd(q, k, v, scale, dropout_p, seed)
        ctx.save_for_backward(x, w, qkv, o, lse, seed, cos_t, sin_t)
        ctx.cfg = (pos0, nh, nkv, scale, dropout_p)
        ctx.has_bias = b is not None
        ctx.w_param = w if sole_use else None
        ctx.b_param = b if sole_use else None
        return o

    @staticmethod
    def backward(ctx, do):
        from .linear import _input_grad, _param_grads
        x, w, qkv, o, lse, seed, cos_t, sin_t = ctx.saved_tensors
        pos0, nh, nkv, scale, dropout_p = ctx.cfg
        q, k, v = split_qkv(qkv, nh, nkv)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = split_qkv(dqkv, nh, nkv)
        ops().attn_bwd(_packed(do), q, k, v, o, lse, dq, dk, dv, scale, dropout_p, seed, cos_t, sin_t, pos0)
        d2 = dqkv.view(-1, dqkv.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = _input_grad(d2, w).view(x.shape) if ctx.needs_input_grad[0] else None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dw, db = _param_grads(ctx.w_param, ctx.b_param, d2, x2, ctx.needs_input_grad[1], need_b)
        return (dx, dw, db) + (None,) * 9


def qkv_rope_attention_supported(x: torch.Tensor, w: torch.Tensor, b, cos_t: torch.Tensor, sin_t: torch.Tensor,
                                 pos0: int, hs: int) -> bool:
    """Shapes the fused projection + rotary + attention path takes: bf16 on the GPU, head size 64 or 128,
    sequences of a multiple of 128 tokens (a 128-row block of the GEMM's output tile must not straddle two
    sequences), tables that cover the positions (and row 16, the step of the epilogue's angle addition)."""
    from .linear import gemm_supported
    if not (use_hip(x, w, cos_t, sin_t) and x.dim() == 3 and x.dtype == torch.bfloat16 and hs in (64, 128)):
        return False
    if os.environ.get("DPA_FUSED_ROPE", "1") == "0" or not hasattr(ops(), "gemm_nt_rope"):
        return False
    T = x.shape[1]
    if T % 128 != 0 or cos_t.dtype != torch.float32 or sin_t.dtype != torch.float32 or cos_t.shape != sin_t.shape:
        return False
    if not (cos_t.dim() == 2 and cos_t.is_contiguous() and sin_t.is_contiguous() and cos_t.shape[1] * 2 == hs
            and cos_t.shape[0] >= max(pos0 + T, 17) and cos_t.data_ptr() % 16 == 0 and sin_t.data_ptr() % 16 == 0):
        return False
    return w.shape[0] % hs == 0 and gemm_supported(x.reshape(-1, x.shape[-1]), w, b)


def qkv_rope_attention(x: torch.Tensor, w: torch.Tensor, b, cos_t: torch.Tensor, sin_t: torch.Tensor, pos0: int,
                       nh: int, nkv: int, dropout_p: float = 0.0, sole_use: bool = False) -> torch.Tensor:
    """``x [B, T, C]`` -> ``o [B, T, nh, hs]`` = causal attention over the rotary-embedded fused projection
    ``x w^T + b`` (``w [(nh + 2 nkv) hs, C]``); callers check ``qkv_rope_attention_supported`` first."""
    hs = w.shape[0] // (nh + 2 * nkv)
    return _QkvRopeAttnFn.apply(x, w, b, cos_t, sin_t, int(pos0), nh, nkv, hs, 1.0 / math.sqrt(hs), float(dropout_p),
                                sole_use)


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
                     dropout_p: float = 0.0, seed: torch.Tensor | None = None) -> torch.Tensor:
    """``seed`` (int64 ``[1]`` on the GPU) pins the dropout mask; by default one is drawn per call."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    dropout_p = float(dropout_p)
    if _hip_ok(q, k, v):
        return _FlashAttnFn.apply(q, k, v, float(scale), dropout_p, seed)
    if _paddable(q, k, v):
        # Other widths run on the flash kernels too, zero-padded to the next pair they are built for: zero
        # features change neither the scores (the scale is explicit) nor the kept output columns, and no
        # T x T matrix is formed.
        hs, dv = q.shape[-1], v.shape[-1]
        tk, tv = next((a, b) for (a, b) in sorted(_WIDTHS, key=lambda w: w[0] * 3 + w[1]) if a >= hs and b >= dv)
        qp, kp = F.pad(q, (0, tk - hs)), F.pad(k, (0, tk - hs))
        vp = F.pad(v, (0, tv - dv))
        return _FlashAttnFn.apply(qp, kp, vp, float(scale), dropout_p, seed)[..., :dv]
    return causal_attention_eager(q, k, v, float(scale), dropout_p)


def cached_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_len: torch.Tensor,
                     scale: float | None = None) -> torch.Tensor:
    """Inference attention over a preallocated cache: ``q [B, Tq, nh, hs]`` are the newest ``Tq`` positions,
    ``k_cache, v_cache [B, S, nkv, hs]`` hold ``kv_len`` (int32 ``[1]`` on the device, new tokens included)
    valid positions. The length is read by the kernel, not by the host, so the call can sit inside a
    captured hipGraph whose replays see a growing cache. No gradient."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if _hip_ok(q, k_cache, v_cache):
        if (q.shape[1] == 1 and q.shape[2] % k_cache.shape[2] == 0 and hasattr(ops(), "attn_decode")
                and os.environ.get("DPA_FUSED_DECODE", "1") != "0"):
            # one query per sequence: the keys are divided over the waves instead of the queries
            return ops().attn_decode(_packed(q), k_cache, v_cache, float(scale), kv_len)
        o, _ = ops().attn_fwd(_packed(q), k_cache, v_cache, float(scale), 0.0, None, kv_len)
        return o
    n = int(kv_len.item())
    return causal_attention_eager(q, k_cache[:, :n], v_cache[:, :n], float(scale))


def kv_append(k_cache: torch.Tensor, v_cache: torch.Tensor, k_new: torch.Tensor, v_new: torch.Tensor,
              pos: torch.Tensor) -> None:
    """Write the new key / value rows ``[B, 1, H, hs]`` at row ``pos`` (int64 ``[1]`` on the device) of the
    preallocated caches ``[B, S, H, hs]``: one launch, position read on the device."""
    ok = (use_hip(k_cache, k_new) and k_cache.dtype == torch.bfloat16 and k_new.dtype == torch.bfloat16
          and k_cache.is_contiguous() and v_cache.is_contiguous() and (k_cache.shape[2] * k_cache.shape[3]) % 8 == 0)
    if ok:
        for t in (k_new, v_new):
            ok = ok and (t.stride(3) == 1 and (t.shape[2] == 1 or t.stride(2) == t.shape[3]) and t.stride(0) % 8 == 0
                         and t.data_ptr() % 16 == 0)
    if ok:
        ops().kv_append(k_cache, v_cache, k_new, v_new, pos)
    else:
        k_cache.index_copy_(1, pos, k_new)
        v_cache.index_copy_(1, pos, v_new)
