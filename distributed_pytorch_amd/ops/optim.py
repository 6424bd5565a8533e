"""AdamW on flat buffers and the squared gradient norm (HIP on the GPU, eager on the CPU).

State layout: three fp32 flat tensors (master weights, exp_avg, exp_avg_sq) covering a
contiguous element range of a flat parameter buffer. ``segments`` is a list of
``(end_offset, weight_decay)`` pairs relative to the start of the range.

Math is that of ``torch.optim.AdamW`` (decoupled decay, bias correction, eps outside the
sqrt), which the reference uses with ``fused=True`` (reference: single-gpu/model.py:619-637).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch

from .._ext import ops, use_hip


def sumsq(g: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """out (1-element fp32) = [out +] sum(g^2)."""
    if use_hip(g, out) and g.numel() % 8 == 0 and g.is_contiguous() and g.data_ptr() % 16 == 0:
        ops().sumsq(g, out, accumulate)
    else:
        s = g.float().pow(2).sum().reshape(1)
        if accumulate:
            out.add_(s)
        else:
            out.copy_(s)
    return out


def adamw_step(master: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, grad: torch.Tensor,
               param_out: torch.Tensor | None, segments: Sequence[tuple[int, float]], lr: float, beta1: float,
               beta2: float, eps: float, step: int, norm2: torch.Tensor | None = None, max_norm: float = 0.0,
               grad_mul: float = 1.0, dyn: torch.Tensor | None = None) -> None:
    """One AdamW step over a flat range.

    ``grad`` may be bf16 or fp32. If ``norm2`` (device scalar holding the squared global
    gradient norm *before* ``grad_mul``) is given, gradients are clipped to ``max_norm`` (when
    > 0) and the whole update is skipped when it is not finite. ``param_out`` receives the
    updated weights in its own dtype (the bf16 compute copy). ``dyn`` (device fp32
    ``[lr, 1 - beta1^t, sqrt(1 - beta2^t)]``) overrides ``lr`` / ``step`` from device memory, which a
    step replayed from a captured graph needs (kernel arguments are frozen at capture)."""
    n = master.numel()
    if n == 0:
        return
    if use_hip(master, grad) and n % 8 == 0:
        ends = [int(e) for e, _ in segments]
        wds = [float(w) for _, w in segments]
        ops().adamw_step(master, exp_avg, exp_avg_sq, grad, param_out, ends, wds, lr, beta1, beta2, eps, step,
                         norm2, max_norm, grad_mul, dyn)
        from .linear import bump_weights_epoch
        bump_weights_epoch()   # the kernel rewrote compute weights behind autograd's back: drop cached transposes
        return
    # ---- eager path (CPU plumbing / tests) ----
    g = grad.float() * grad_mul
    if norm2 is not None:
        n2 = float(norm2)
        if not math.isfinite(n2):
            return
        if max_norm > 0:
            clip = max_norm / (math.sqrt(n2) * abs(grad_mul) + 1e-6)
            if clip < 1.0:
                g = g * clip
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2s = math.sqrt(1 - beta2 ** step)
    if dyn is not None:
        lr, bc1, bc2s = (float(v) for v in dyn.tolist())
    start = 0
    for end, wd in segments:
        sl = slice(start, end)
        if wd != 0.0:
            master[sl].mul_(1 - lr * wd)
        start = end
    denom = exp_avg_sq.sqrt().div_(bc2s).add_(eps)
    master.addcdiv_(exp_avg, denom, value=-lr / bc1)
    if param_out is not None:
        param_out.copy_(master)
