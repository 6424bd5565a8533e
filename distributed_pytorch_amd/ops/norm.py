"""LayerNorm with optional fused residual add (HIP on the GPU, eager on the CPU).

``layer_norm(x, w, b)``                -> y
``add_layer_norm(x, r, w, b)``         -> (x + r, LN(x + r))   one pass over memory
``layer_norm_keep(x, w, b)``           -> (x, LN(x))           x's other consumers send their gradient here

Reference behaviour: ``nn.LayerNorm(C)`` with eps 1e-5, affine weight and bias
(reference: single-gpu/model.py:514-515,557).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip

_MAX_HIP_COLS = 4096


def _hip_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    C = x.shape[-1]
    return (use_hip(x, w) and C % 8 == 0 and C <= _MAX_HIP_COLS and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float32))


def _direct_targets(w, b):
    """(w.grad, b.grad) when the owning engine asked for direct accumulation (``_dpa_direct``) and both
    are flat-buffer views the kernel can add into, else None."""
    if not (getattr(w, "_dpa_direct", False) and getattr(b, "_dpa_direct", False)):
        return None
    gw, gb = w.grad, b.grad
    for g, p in ((gw, w), (gb, b)):
        if g is None or g.dtype != p.dtype or not g.is_contiguous() or g.shape != p.shape:
            return None
    return gw, gb


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, w, b, eps, keep=False):
        ctx.params = (w, b)
        xc = x.contiguous()
        rc = res.contiguous() if res is not None else None
        y, s, mean, rstd = ops().layernorm_fwd(xc, rc, w, b, eps)
        ln_in = s if res is not None else xc
        ctx.save_for_backward(ln_in, w, mean, rstd)
        ctx.has_res = res is not None
        ctx.keep = keep
        if res is not None:
            return s, y
        if keep:
            # the input handed back as a second output: whatever else consumes it downstream sends its
            # gradient through this node, where it is added inside the backward kernel
            return x, y
        return y

    @staticmethod
    def backward(ctx, *grads):
        ln_in, w, mean, rstd = ctx.saved_tensors
        if ctx.has_res:
            ds, dy = grads
            if dy is None:
                return ds, ds, None, None, None, None
            dres = ds.contiguous() if ds is not None else None
            dx, dw, db = _ln_backward(ctx, dy.contiguous(), ln_in, w, mean, rstd, dres)
            return dx, dx, dw, db, None, None
        if ctx.keep:
            ds, dy = grads
            if dy is None:
                return ds, None, None, None, None, None
            dres = ds.contiguous() if ds is not None else None
            dx, dw, db = _ln_backward(ctx, dy.contiguous(), ln_in, w, mean, rstd, dres)
            return dx, None, dw, db, None, None
        (dy,) = grads
        dx, dw, db = _ln_backward(ctx, dy.contiguous(), ln_in, w, mean, rstd, None)
        return dx, None, dw, db, None, None


def _ln_backward(ctx, dy, ln_in, w, mean, rstd, dres):
    """dx plus the parameter gradients, which are added straight into the engine's flat gradient
    buffer when the parameters ask for it (None is then returned for them; their post-accumulate
    hooks fire all the same)."""
    tgt = _direct_targets(*ctx.params) if ctx.needs_input_grad[2] and ctx.needs_input_grad[3] else None
    if tgt is not None:
        return ops().layernorm_bwd_acc(dy, ln_in, w, mean, rstd, dres, tgt[0], tgt[1]), None, None
    return ops().layernorm_bwd(dy, ln_in, w, mean, rstd, dres)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    if _hip_ok(x, w):
        return _LayerNormFn.apply(x, None, w, b, eps)
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def layer_norm_keep(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """Returns ``(x, layer_norm(x))``. Use the returned ``x`` for every further consumer of the input (a
    residual connection): its gradient then arrives at this node and is added to the LayerNorm's input
    gradient inside the backward kernel instead of by a separate accumulation pass."""
    if _hip_ok(x, w):
        return _LayerNormFn.apply(x, None, w, b, eps, True)
    return x, F.layer_norm(x, (x.shape[-1],), w, b, eps)


def add_layer_norm(x: torch.Tensor, r: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """Returns ``(x + r, layer_norm(x + r))``."""
    if _hip_ok(x, w) and r.dtype == x.dtype:
        return _LayerNormFn.apply(x, r, w, b, eps)
    s = x + r
    return s, F.layer_norm(s, (s.shape[-1],), w, b, eps)
