"""Feed-forward activations. Every name of the reference's activation map has a HIP
forward / backward kernel on the GPU (GELU exact-erf and SwiGLU their own, the rest one kernel
templated on the function) and the stock op on the CPU (reference: single-gpu/model.py:376-394).
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


# kernel index (csrc/activation.hip::ActKind) and eager form of the other activations
_KINDS = {
    "relu": (0, F.relu), "silu": (1, F.silu), "swish": (1, F.silu), "mish": (2, F.mish), "selu": (3, F.selu),
    "elu": (4, F.elu), "celu": (4, F.celu), "sigmoid": (5, torch.sigmoid), "tanh": (6, torch.tanh),
    "lrelu": (7, lambda x: F.leaky_relu(x, 0.01)),
}


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.kind = kind
        return ops().act_fwd(xc, kind)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops().act_bwd(dy.contiguous(), x, ctx.kind), None


def _elementwise(name: str):
    kind, eager = _KINDS[name]

    def fn(x: torch.Tensor) -> torch.Tensor:
        if _hip_ok(x, 8):
            return _ActFn.apply(x, kind)
        return eager(x)

    fn.__name__ = name
    return fn


_ELEMENTWISE = {name: _elementwise(name) for name in _KINDS}
_ELEMENTWISE["gelu"] = gelu


def get_activation(name: str):
    """Activation by reference name; unknown names fall back to GELU like the reference's
    ``non_linearity_map.get(name, nn.GELU())``. ``glu`` halves the width (and therefore cannot
    feed ``c_proj`` of width ``up_dim``, exactly as in the reference)."""
    name = name.lower()
    if name == "glu":
        return lambda x: F.glu(x, dim=-1)
    return _ELEMENTWISE.get(name, gelu)
