"""Linear layer with a hand-written weight-gradient path.

Forward ``y = x @ w.T + b`` and the input gradient ``dx = dy @ w`` are plain library GEMMs
(hipBLASLt through ``torch``): both contract over a contiguous dimension and run near the
library's best. The weight gradient ``dw = dy.T @ x`` contracts over the tokens with both
operands strided; it runs on the split-M MFMA kernel of ``csrc/gemm_wgrad.hip`` (2-3x the
library on these shapes), which also produces the bias gradient from the tiles it already
stages instead of a separate column-sum reduction.

Reference behaviour: ``nn.Linear`` (reference: single-gpu/model.py:112-114,373-384).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip


def wgrad_supported(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    return (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16
            and dy2.shape[1] % 128 == 0 and x2.shape[1] % 128 == 0 and dy2.shape[0] >= 256
            and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0
            and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0)


def wgrad2_supported(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    """Shapes of the LDS-DMA kernel: tokens % 64 == 0, K % 256 == 0."""
    return (dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16
            and dy2.shape[0] % 64 == 0 and dy2.shape[0] >= 256 and dy2.shape[1] % 8 == 0 and x2.shape[1] % 256 == 0
            and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 8 == 0 and x2.stride(0) % 8 == 0
            and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0)


# Output sizes from which the library's split-K kernels overtake the hand-written weight-gradient
# kernels (measured on MI355X, benchmarks/wgrad_shapes_bench.py: own kernels win up to 4.2 M output
# elements, the library from 12.6 M; a 256x256 tiling of a large output no longer divides evenly
# over 256 CUs and the split-M advantage is gone).
LIBRARY_WGRAD_MIN_OUT = int(os.environ.get("DPA_LIBRARY_WGRAD_MIN_OUT", 8 * 1024 * 1024))


def _library_wgrad(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    return dy2.shape[1] * x2.shape[1] >= LIBRARY_WGRAD_MIN_OUT


def bias_grad(dy2: torch.Tensor) -> torch.Tensor:
    """Column sums of dy2 [M, N] in its own dtype (fp32 accumulation on the GPU kernel)."""
    if (dy2.is_cuda and dy2.dtype in (torch.bfloat16, torch.float32) and dy2.stride(1) == 1
            and dy2.shape[1] % 8 == 0 and dy2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0):
        return ops().colsum(dy2).to(dy2.dtype)
    return dy2.sum(0)


def weight_grad(dy2: torch.Tensor, x2: torch.Tensor, want_bias: bool, version: int = 0):
    """(dw [N, K], dbias [N] | None) for dy2 [M, N], x2 [M, K] in bf16. ``version``: 0 = fastest path for
    the shape (library for very large outputs), 1 = register-staged kernel, 2 = LDS-DMA kernels."""
    N, K = dy2.shape[1], x2.shape[1]
    if version == 0 and _library_wgrad(dy2, x2):
        return torch.mm(dy2.t(), x2), (bias_grad(dy2) if want_bias else None)
    if version != 1 and wgrad2_supported(dy2, x2):
        if want_bias:
            slabs, db = ops().wgrad2_slabs_bias(dy2, x2)   # bias sums come out of the same kernel
            dw = torch.empty(N, K, dtype=dy2.dtype, device=dy2.device)
            ops().ws_to(slabs, dw, False)
            return dw, db.to(dy2.dtype)
        return ops().wgrad2(dy2, x2), None
    ws = torch.zeros(N, K, dtype=torch.float32, device=dy2.device)
    bws = torch.zeros(N, dtype=torch.float32, device=dy2.device) if want_bias else None
    ops().wgrad_into(dy2, x2, ws, bws)
    dw = torch.empty(N, K, dtype=dy2.dtype, device=dy2.device)
    ops().ws_to(ws, dw, False)
    return dw, (bws.to(dy2.dtype) if want_bias else None)


def _direct_target(w_param, dy2, x2):
    """The flat-buffer gradient view of ``w_param`` if the engine that owns it asked for direct
    accumulation (``_dpa_direct``) and the kernel can write it, else None."""
    if not getattr(w_param, "_dpa_direct", False):
        return None
    g = w_param.grad
    if g is None or g.dtype not in (torch.bfloat16, torch.float32) or not g.is_contiguous() \
            or g.data_ptr() % 16 != 0 or g.numel() % 8 != 0:
        return None
    return g if (wgrad2_supported(dy2, x2) and not _library_wgrad(dy2, x2)) else None


def _direct_bias_target(b_param):
    """The flat-buffer gradient view of a bias that asked for direct accumulation, else None."""
    if b_param is None or not getattr(b_param, "_dpa_direct", False):
        return None
    g = b_param.grad
    if g is None or g.dtype not in (torch.bfloat16, torch.float32) or not g.is_contiguous() \
            or g.data_ptr() % 16 != 0 or g.numel() % 8 != 0:
        return None
    return g


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, sole_use):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.w_param = w if sole_use else None
        ctx.b_param = b if sole_use else None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, w).view(x.shape)
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        target = None
        if ctx.w_param is not None and ctx.needs_input_grad[1]:
            target = _direct_target(ctx.w_param, dy2, x2)
        if target is not None:
            # Fold the split slabs straight into the engine's flat gradient buffer (one pass instead
            # of convert + autograd's accumulate). Returning None for dw still runs the parameter's
            # AccumulateGrad node, so post-accumulate hooks (bucket / unit accounting of the
            # engines) fire exactly as they do for an autograd-accumulated gradient.
            btarget = _direct_bias_target(ctx.b_param) if need_b else None
            if need_b and btarget is None:
                slabs, dbf = ops().wgrad2_slabs_bias(dy2, x2)
                db = dbf.to(dy2.dtype)
                ops().ws_to(slabs, target, True)
            else:
                # GEMM + folds of the weight and bias slabs into the flat buffers: one op, no temporaries
                ops().wgrad2_direct(dy2, x2, target, btarget)
        elif ctx.needs_input_grad[1]:
            if wgrad2_supported(dy2, x2) or wgrad_supported(dy2, x2):
                dw, db = weight_grad(dy2, x2, need_b)
            else:
                dw = torch.mm(dy2.t(), x2)
                if need_b:
                    db = bias_grad(dy2)
        elif need_b:
            db = bias_grad(dy2)
        return dx, dw, db, None


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None, sole_use: bool = False) -> torch.Tensor:
    """``x @ w.T + b``. ``sole_use=True`` promises that this call is the only consumer of ``w`` in
    the forward pass, which allows its gradient to be accumulated straight into the owning
    engine's flat gradient buffer (see ``_direct_target``)."""
    if use_hip(x, w) and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and torch.is_grad_enabled() \
            and (x.requires_grad or w.requires_grad):
        return _LinearFn.apply(x, w, b, sole_use)
    return F.linear(x, w, b)
