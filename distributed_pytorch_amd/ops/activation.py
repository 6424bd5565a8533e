"""Feed-forward activations. GELU (exact erf) and SwiGLU have HIP forward/backward kernels;
the remaining names of the reference's activation map run as stock elementwise ops
(reference: single-gpu/model.py:376-394).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        return ops().gelu_fwd(xc)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops().gelu_bwd(dy.contiguous(), x)


class _SwigluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        hc = h.contiguous()
        ctx.save_for_backward(hc)
        return ops().swiglu_fwd(hc)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return ops().swiglu_bwd(dy.contiguous(), h)


def _hip_ok(x: torch.Tensor, mult: int) -> bool:
    return use_hip(x) and x.dtype in (torch.bfloat16, torch.float32) and x.shape[-1] % mult == 0


def gelu(x: torch.Tensor) -> torch.Tensor:
    if _hip_ok(x, 8):
        return _GeluFn.apply(x)
    return F.gelu(x)


def swiglu(h: torch.Tensor) -> torch.Tensor:
    """``silu(h[..., :U]) * h[..., U:]`` for ``h [..., 2U]``."""
    if _hip_ok(h, 16):
        return _SwigluFn.apply(h)
    a, b = h.chunk(2, dim=-1)
    return F.silu(a) * b


_ELEMENTWISE = {
    "relu": F.relu, "gelu": gelu, "swish": F.silu, "silu": F.silu, "mish": F.mish, "selu": F.selu,
    "celu": F.celu, "elu": F.elu, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
    "lrelu": lambda x: F.leaky_relu(x, 0.01),
}


def get_activation(name: str):
    """Activation by reference name; unknown names fall back to GELU like the reference's
    ``non_linearity_map.get(name, nn.GELU())``. ``glu`` halves the width (and therefore cannot
    feed ``c_proj`` of width ``up_dim``, exactly as in the reference)."""
    name = name.lower()
    if name == "glu":
        return lambda x: F.glu(x, dim=-1)
    return _ELEMENTWISE.get(name, gelu)
