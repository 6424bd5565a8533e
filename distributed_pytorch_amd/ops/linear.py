"""Linear layer on hand-written GEMMs.

Forward ``y = x @ w.T + b`` and the input gradient ``dx = dy @ w`` run on the NT MFMA kernel of
``csrc/gemm_nt.hip`` (bias, GELU / GELU', multiply and add epilogues; the input gradient uses the
weight transposed once per optimizer step); shapes outside its envelope (fewer than
``GEMM_MIN_ROWS`` rows, K not a multiple of 64, ...) fall back to the library. The weight gradient ``dw = dy.T @ x`` contracts over the tokens with both
operands strided; it runs on the split-M MFMA kernel of ``csrc/gemm_wgrad.hip`` (2-3x the
library on these shapes), which also produces the bias gradient from the tiles it already
stages instead of a separate column-sum reduction.

Reference behaviour: ``nn.Linear`` (reference: single-gpu/model.py:112-114,373-384).
"""
from __future__ import annotations

import collections
import os
import weakref

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


def _library_wgrad(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    """Weight gradients run on the own kernels at every size (the 256 x 256 kernel numbers its tiles in blocks per
    XCD and writes a one-split product straight into the gradient: benchmarks/wgrad_shapes_bench.py); the library
    serves them only under ``DPA_LIBRARY_GEMM=1``, the switch of the A/B comparisons."""
    return os.environ.get("DPA_LIBRARY_GEMM", "0") == "1"


def bias_grad(dy2: torch.Tensor) -> torch.Tensor:
    """Column sums of dy2 [M, N] in its own dtype (fp32 accumulation on the GPU kernel)."""
    if (dy2.is_cuda and dy2.dtype in (torch.bfloat16, torch.float32) and dy2.stride(1) == 1
            and dy2.shape[1] % 8 == 0 and dy2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0):
        return ops().colsum(dy2).to(dy2.dtype)
    return dy2.sum(0)


def weight_grad(dy2: torch.Tensor, x2: torch.Tensor, want_bias: bool, version: int = 0):
    """(dw [N, K], dbias [N] | None) for dy2 [M, N], x2 [M, K] in bf16. ``version``: 0 = the kernel
    picked for the shape, 1 = register-staged kernel, 2 = LDS-DMA kernels."""
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
    g = getattr(w_param, "_dpa_grad32", None)   # the engine's fp32 gradient buffer, if it keeps one
    if g is None:
        g = w_param.grad
    if g is None or g.dtype not in (torch.bfloat16, torch.float32) or not g.is_contiguous() \
            or g.data_ptr() % 16 != 0 or g.numel() % 8 != 0:
        return None
    return g if (wgrad2_supported(dy2, x2) and not _library_wgrad(dy2, x2)) else None


def _direct_bias_target(b_param):
    """The flat-buffer gradient view of a bias that asked for direct accumulation, else None."""
    if b_param is None or not getattr(b_param, "_dpa_direct", False):
        return None
    g = getattr(b_param, "_dpa_grad32", None)
    if g is None:
        g = b_param.grad
    if g is None or g.dtype not in (torch.bfloat16, torch.float32) or not g.is_contiguous() \
            or g.data_ptr() % 16 != 0 or g.numel() % 8 != 0:
        return None
    return g


# ---- own NT GEMM (csrc/gemm_nt.hip) -------------------------------------------------------------
# Forward y = x W^T and input gradient dx = dy W both run on the same kernel (both operands
# contraction-contiguous): the input gradient takes W^T, transposed once per weight version.
GEMM_MIN_ROWS = int(os.environ.get("DPA_GEMM_MIN_ROWS", 512))   # below this the 256-row tiles are mostly padding
_EPI_NONE, _EPI_GELU, _EPI_MUL, _EPI_ADD, _EPI_SWIGLU, _EPI_DSWIGLU = 0, 1, 2, 3, 4, 5
_WT_CACHE: dict = {}
_BATCH_TRANSPOSES = os.environ.get("DPA_BATCH_TRANSPOSES", "1") == "1"
_WEIGHTS_EPOCH = 0


def bump_weights_epoch() -> None:
    """Called after every in-place parameter update that bypasses autograd's version counter (the
    fused optimizer kernel): cached transposed weights are stale from here on."""
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


_PTR_TABLES: dict = {}


def _refresh_resident_transposes() -> None:
    """One launch per weight shape for every cached transpose that went stale with the last optimizer step.

    Only weights whose owner keeps them at a fixed address for the whole run take part (``Parameter._dpa_resident``,
    set by the flat groups of the data-parallel engines; the units of the fully sharded engine release and re-gather
    their storage, so a stale entry of theirs may have nothing valid behind it right now). The transposes are written
    into the buffers the cache already holds, whose addresses stay in a device table per shape."""
    by_shape: dict = {}
    for key, (stamp, wt, ref) in _WT_CACHE.items():
        w = ref()
        if w is None or not getattr(w, "_dpa_resident", False) or w.dtype != torch.bfloat16 or not w.is_contiguous():
            continue
        now = (_WEIGHTS_EPOCH, w._version, w.data_ptr())
        if stamp == now or stamp[2] != now[2] or wt.shape != (w.shape[1], w.shape[0]):
            continue
        by_shape.setdefault((w.shape[0], w.shape[1], w.device), []).append((key, w, wt, ref, now))
    for (R, C, dev), items in by_shape.items():
        if R % 8 or C % 8:
            continue
        ptrs = tuple((w.data_ptr(), wt.data_ptr()) for _, w, wt, _, _ in items)
        tab = _PTR_TABLES.get(ptrs)
        if tab is None:
            if len(_PTR_TABLES) > 64:
                _PTR_TABLES.clear()
            tab = _PTR_TABLES[ptrs] = (torch.tensor([a for a, _ in ptrs], dtype=torch.int64, device=dev),
                                       torch.tensor([b for _, b in ptrs], dtype=torch.int64, device=dev))
        ops().transpose_bf16_many(tab[0], tab[1], R, C)
        for key, w, wt, ref, now in items:
            _WT_CACHE[key] = (now, wt, ref)


def transposed_weight(w: torch.Tensor) -> torch.Tensor:
    """``w.T`` as a contiguous bf16 matrix, cached per weight version (recomputed inside a hipGraph
    capture, where the replay has to redo it)."""
    if torch.cuda.is_current_stream_capturing():
        # the warm-up steps left a buffer for this weight: refill it, so that its address (which grouped products
        # keep in cached pointer tables) is the same in every replay and outside the graph's own pool
        ent = _WT_CACHE.get(id(w))
        if ent is not None and ent[2]() is w and ent[1].shape == (w.shape[1], w.shape[0]):
            ent[1].copy_(w.detach().t())
            return ent[1]
        return ops().transpose_bf16(w.detach())
    # Keyed by the weight tensor itself, not by its address: engines that gather parameters into
    # recycled buffers (fully sharded units) put different weights of equal shape at the same address.
    key = id(w)
    stamp = (_WEIGHTS_EPOCH, w._version, w.data_ptr())
    ent = _WT_CACHE.get(key)
    if ent is not None and ent[0] == stamp and ent[2]() is w:
        return ent[1]
    if ent is not None and ent[2]() is w and getattr(w, "_dpa_resident", False) and _BATCH_TRANSPOSES:
        # the first stale request after an optimizer step refreshes every resident weight's transpose
        _refresh_resident_transposes()
        ent = _WT_CACHE.get(key)
        if ent is not None and ent[0] == stamp:
            return ent[1]
    wt = ops().transpose_bf16(w.detach())
    if len(_WT_CACHE) > 4096:
        _WT_CACHE.clear()
    # The entry leaves with its weight: a weight that is a fresh tensor object every step (a view handed out by an
    # autograd node, e.g. the diverted head weight of the split tied tail) would otherwise leave one transposed copy
    # behind per step.
    _WT_CACHE[key] = (stamp, wt, weakref.ref(w, lambda _r, k=key: _drop_dead_transpose(k)))
    return wt


def _drop_dead_transpose(key) -> None:
    ent = _WT_CACHE.get(key)
    if ent is not None and ent[2]() is None:
        del _WT_CACHE[key]


# 256 x 256 output tiles below which the library serves a product: half a round of the 256 workgroups. GPT-2-124M at
# micro-batch 4 / 8 (4096 / 8192 rows): +6 % / +3 % tokens/s over running every product on the own kernel; from
# micro-batch 16 on every layer product is above the bar.
GEMM_MIN_TILES = int(os.environ.get("DPA_GEMM_MIN_TILES", 128))


def _rows_ok(x2: torch.Tensor, n_out: int = 1 << 30) -> bool:
    tiles = ((x2.shape[0] + 255) // 256) * ((n_out + 255) // 256)
    return x2.shape[0] >= GEMM_MIN_ROWS and x2.stride(1) == 1 and tiles >= GEMM_MIN_TILES


def gemm_supported(x2: torch.Tensor, w: torch.Tensor, b=None, aux=None) -> bool:
    return (os.environ.get("DPA_LIBRARY_GEMM", "0") != "1" and x2.is_cuda and x2.dtype == torch.bfloat16
            and w.dtype == torch.bfloat16 and x2.dim() == 2 and _rows_ok(x2, w.shape[0]) and w.is_contiguous()
            and ops().gemm_nt_supported(x2, w, b, aux))


def _dx_supported(dy2: torch.Tensor, w: torch.Tensor, aux=None) -> bool:
    """dx = dy2 [M, N] @ w [N, K] through the NT kernel on w^T [K, N]."""
    N, K = w.shape
    if not (os.environ.get("DPA_LIBRARY_GEMM", "0") != "1" and dy2.is_cuda and dy2.dtype == torch.bfloat16
            and w.dtype == torch.bfloat16 and w.is_contiguous() and _rows_ok(dy2, K) and N % 64 == 0 and N >= 128
            and K % 16 == 0 and dy2.stride(0) % 8 == 0 and dy2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0):
        return False
    return aux is None or (aux.stride(1) == 1 and aux.stride(0) % 8 == 0 and aux.data_ptr() % 16 == 0)


def _input_grad(dy2: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    if _dx_supported(dy2, w):
        return ops().gemm_nt(dy2, transposed_weight(w), None, _EPI_NONE, None)[0]
    return torch.mm(dy2, w)


# ---- deferred weight gradients --------------------------------------------------------------------------------
# Expert-parallel layers want the input gradient of the local experts on the wire before the experts' weight
# gradients are computed: the exchange of the input gradients then travels while those GEMMs run. Parameters marked
# ``_dpa_defer`` (local experts of the expert-parallel engines: never reduced across ranks, accumulated straight
# into the flat buffers, no bucket accounting waits for them) have their weight-gradient products queued while a
# deferral window is open; ``flush_deferred_wgrads`` runs them (parallel/ep.py opens the window right before the
# experts' backward and flushes it once the exchange of their input gradient has been issued; the engines flush
# again before they read any gradient, so nothing is ever left behind).
class _Deferral:
    active = False
    queue: list = []
    log = collections.deque(maxlen=1024)      # ("defer" | "run", n): the most recent events, what tests look at


def begin_deferred_wgrads() -> None:
    _Deferral.active = True


def flush_deferred_wgrads() -> int:
    """Run the queued weight-gradient products and close the window; returns how many ran."""
    _Deferral.active = False
    q, _Deferral.queue = _Deferral.queue, []
    for fn in q:
        fn()
    if q:
        _Deferral.log.append(("run", len(q)))
    return len(q)


def maybe_defer_wgrad(params, fn) -> bool:
    """Queue ``fn`` (a closure that computes and accumulates the weight gradients of ``params`` into their flat
    buffers) if a deferral window is open and every parameter allows it; False: the caller runs it now."""
    if not _Deferral.active or not all(p is not None and getattr(p, "_dpa_defer", False) for p in params):
        return False
    _Deferral.queue.append(fn)
    _Deferral.log.append(("defer", len(params)))
    return True


def _param_grads(w_param, b_param, dy2, x2, need_w: bool, need_b: bool):
    """(dw, db) of a Linear for autograd to accumulate, or (None, None) after folding them straight
    into the owning engine's flat gradient buffers (``w_param`` / ``b_param`` given and marked
    ``_dpa_direct``). Returning None for a gradient still runs the parameter's AccumulateGrad node, so
    post-accumulate hooks (bucket / unit accounting of the engines) fire exactly as for an
    autograd-accumulated gradient."""
    dw = db = None
    target = _direct_target(w_param, dy2, x2) if (w_param is not None and need_w) else None
    if target is not None:
        btarget = _direct_bias_target(b_param) if need_b else None
        if not need_b and maybe_defer_wgrad([w_param], lambda: ops().wgrad2_direct(dy2, x2, target, None)):
            return None, None
        if need_b and btarget is None:
            slabs, dbf = ops().wgrad2_slabs_bias(dy2, x2)
            db = dbf.to(dy2.dtype)
            ops().ws_to(slabs, target, True)
        else:
            # GEMM + folds of the weight and bias slabs into the flat buffers: one op, no temporaries
            ops().wgrad2_direct(dy2, x2, target, btarget)
    elif need_w:
        if wgrad2_supported(dy2, x2) or wgrad_supported(dy2, x2):
            dw, db = weight_grad(dy2, x2, need_b)
        else:
            dw = torch.mm(dy2.t(), x2)
            if need_b:
                db = bias_grad(dy2)
    elif need_b:
        db = bias_grad(dy2)
    return dw, db


def _residual_rows(residual, w):
    """The residual as [rows, N] rows the add epilogue can read, or None."""
    if residual is None or residual.dtype != torch.bfloat16 or residual.shape[-1] != w.shape[0]:
        return None
    r2 = residual.reshape(-1, residual.shape[-1])
    return r2 if (r2.stride(1) == 1 and r2.stride(0) % 8 == 0 and r2.data_ptr() % 16 == 0) else None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, sole_use, residual=None):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.w_param = w if sole_use else None
        ctx.b_param = b if sole_use else None
        x2 = x.reshape(-1, x.shape[-1])
        if residual is not None:
            # y = x W^T + b + residual: the add is the GEMM's epilogue where the kernel takes the shape
            r2 = _residual_rows(residual, w)
            if r2 is not None and r2.shape[0] == x2.shape[0] and gemm_supported(x2, w, b, r2):
                y = ops().gemm_nt(x2, w, b, _EPI_ADD, r2)[0]
                return torch.ops.aten._unsafe_view(y, (*x.shape[:-1], w.shape[0]))
            y = ops().gemm_nt(x2, w, b, _EPI_NONE, None)[0].view(*x.shape[:-1], w.shape[0]) if gemm_supported(x2, w, b) \
                else F.linear(x, w, b)
            return y + residual
        if gemm_supported(x2, w, b):
            # _unsafe_view: the freshly made GEMM output reshaped WITHOUT view tracking, so that the caller
            # may update it in place (rotary embedding of the q / k columns) like any tensor it owns
            return torch.ops.aten._unsafe_view(ops().gemm_nt(x2, w, b, _EPI_NONE, None)[0], (*x.shape[:-1], w.shape[0]))
        # library product on the 2-D rows, reshaped the same way (F.linear on a 3-D input hands back a tracked view,
        # which a custom Function's output must not be if the caller is to update it in place)
        return torch.ops.aten._unsafe_view(F.linear(x2, w, b), (*x.shape[:-1], w.shape[0]))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy2, w).view(x.shape)
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dw, db = _param_grads(ctx.w_param, ctx.b_param, dy2, x2, ctx.needs_input_grad[1], need_b)
        # the residual's gradient is dy itself
        return dx, dw, db, None, (dy if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4] else None)


def _second_product(act, w_proj, residual):
    """c_proj of the fused feed-forward blocks: act W_proj^T (+ residual in the GEMM's epilogue)."""
    if residual is not None:
        r2 = _residual_rows(residual, w_proj)
        if r2 is not None and r2.shape[0] == act.shape[0] and ops().gemm_nt_supported(act, w_proj, None, r2):
            return ops().gemm_nt(act, w_proj, None, _EPI_ADD, r2)[0]
        return ops().gemm_nt(act, w_proj, None, _EPI_NONE, None)[0] + residual.reshape(act.shape[0], -1)
    return ops().gemm_nt(act, w_proj, None, _EPI_NONE, None)[0]


class _MlpGeluFn(torch.autograd.Function):
    """``c_proj(gelu(c_fc(x)))`` for the bias-free feed-forward block (reference:
    single-gpu/model.py:388-398) with the activation inside the GEMM epilogues: the c_fc product
    writes gelu(h) and gelu'(h) directly (the pre-activation h is never stored), and in backward the
    input gradient of c_proj comes out of its GEMM already multiplied by gelu'(h)."""

    @staticmethod
    def forward(ctx, x, w_fc, w_proj, sole_use, residual=None):
        x2 = x.reshape(-1, x.shape[-1])
        act, dact = ops().gemm_nt(x2, w_fc, None, _EPI_GELU, None)
        y = _second_product(act, w_proj, residual)
        ctx.save_for_backward(x, w_fc, w_proj, act, dact)
        ctx.fc_param = w_fc if sole_use else None
        ctx.proj_param = w_proj if sole_use else None
        return torch.ops.aten._unsafe_view(y, (*x.shape[:-1], w_proj.shape[0]))

    @staticmethod
    def backward(ctx, dy):
        x, w_fc, w_proj, act, dact = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        # dh = (dy W_proj) * gelu'(h): the multiply is the GEMM's epilogue
        dh = ops().gemm_nt(dy2, transposed_weight(w_proj), None, _EPI_MUL, dact)[0]
        dw_proj, _ = _param_grads(ctx.proj_param, None, dy2, act, ctx.needs_input_grad[2], False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dh, w_fc).view(x.shape)
        dw_fc, _ = _param_grads(ctx.fc_param, None, dh, x2, ctx.needs_input_grad[1], False)
        return dx, dw_fc, dw_proj, None, (dy if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4] else None)


class _MlpSwigluFn(torch.autograd.Function):
    """``c_proj(silu(gate) * up)`` with ``[gate | up] = c_fc(x)`` (reference: single-gpu/model.py:388-398,
    SwiGLU branch) with the activation inside the GEMM epilogues: the c_fc product pairs every gate column
    with its up column inside one output tile and writes the activation and the pre-activation; in backward
    the input gradient of c_proj leaves its GEMM already turned into ``[dgate | dup]``."""

    @staticmethod
    def forward(ctx, x, w_fc, w_proj, sole_use, residual=None):
        x2 = x.reshape(-1, x.shape[-1])
        act, h = ops().gemm_nt(x2, w_fc, None, _EPI_SWIGLU, None)
        y = _second_product(act, w_proj, residual)
        ctx.save_for_backward(x, w_fc, w_proj, act, h)
        ctx.fc_param = w_fc if sole_use else None
        ctx.proj_param = w_proj if sole_use else None
        return torch.ops.aten._unsafe_view(y, (*x.shape[:-1], w_proj.shape[0]))

    @staticmethod
    def backward(ctx, dy):
        x, w_fc, w_proj, act, h = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dh = ops().gemm_nt(dy2, transposed_weight(w_proj), None, _EPI_DSWIGLU, h)[0]   # [M, 2U] = [dgate | dup]
        dw_proj, _ = _param_grads(ctx.proj_param, None, dy2, act, ctx.needs_input_grad[2], False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dh, w_fc).view(x.shape)
        dw_fc, _ = _param_grads(ctx.fc_param, None, dh, x2, ctx.needs_input_grad[1], False)
        return dx, dw_fc, dw_proj, None, (dy if len(ctx.needs_input_grad) > 4 and ctx.needs_input_grad[4] else None)


def mlp_swiglu_supported(x: torch.Tensor, w_fc: torch.Tensor, w_proj: torch.Tensor) -> bool:
    if not (use_hip(x, w_fc, w_proj) and x.dtype == torch.bfloat16 and w_fc.dtype == torch.bfloat16
            and w_proj.dtype == torch.bfloat16 and hasattr(ops(), "gemm_nt_pair_supported")):
        return False
    if os.environ.get("DPA_LIBRARY_GEMM", "0") == "1" or os.environ.get("DPA_FUSED_SWIGLU", "1") == "0":
        return False
    x2 = x.reshape(-1, x.shape[-1])
    two_up, C = w_fc.shape
    up = two_up // 2
    if x2.shape[0] < GEMM_MIN_ROWS or two_up % 2 or tuple(w_proj.shape) != (C, up) or not w_proj.is_contiguous():
        return False
    if not (up % 64 == 0 and up >= 128 and C % 64 == 0 and C >= 128):
        return False
    return bool(ops().gemm_nt_pair_supported(x2, w_fc, _EPI_SWIGLU, None))


def mlp_swiglu(x: torch.Tensor, w_fc: torch.Tensor, w_proj: torch.Tensor, sole_use: bool = False,
               residual: torch.Tensor | None = None) -> torch.Tensor:
    """Fused SwiGLU feed-forward block (``+ residual`` in the second GEMM's epilogue); callers check
    ``mlp_swiglu_supported`` first."""
    return _MlpSwigluFn.apply(x, w_fc, w_proj, sole_use, residual)


def mlp_gelu_supported(x: torch.Tensor, w_fc: torch.Tensor, w_proj: torch.Tensor) -> bool:
    if not (use_hip(x, w_fc, w_proj) and x.dtype == torch.bfloat16 and w_fc.dtype == torch.bfloat16
            and w_proj.dtype == torch.bfloat16):
        return False
    x2 = x.reshape(-1, x.shape[-1])
    up, C = w_fc.shape
    if not gemm_supported(x2, w_fc):
        return False
    # shapes of the second product and of the two input-gradient products
    return (up % 64 == 0 and up >= 128 and C % 64 == 0 and C >= 128 and w_proj.is_contiguous()
            and tuple(w_proj.shape) == (C, up))


def mlp_gelu(x: torch.Tensor, w_fc: torch.Tensor, w_proj: torch.Tensor, sole_use: bool = False,
             residual: torch.Tensor | None = None) -> torch.Tensor:
    """Fused feed-forward block with exact-erf GELU (``+ residual`` in the second GEMM's epilogue); callers
    check ``mlp_gelu_supported`` first."""
    return _MlpGeluFn.apply(x, w_fc, w_proj, sole_use, residual)


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None = None, sole_use: bool = False,
           residual: torch.Tensor | None = None) -> torch.Tensor:
    """``x @ w.T + b (+ residual)``. ``sole_use=True`` promises that this call is the only consumer of
    ``w`` in the forward pass, which allows its gradient to be accumulated straight into the owning
    engine's flat gradient buffer (see ``_direct_target``). ``residual`` (same shape as the result) is
    added in the GEMM's epilogue on the GPU."""
    if use_hip(x, w) and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16:
        if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (residual is not None and residual.requires_grad)):
            return _LinearFn.apply(x, w, b, sole_use, residual)
        x2 = x.reshape(-1, x.shape[-1])
        if residual is not None:
            r2 = _residual_rows(residual, w)
            if r2 is not None and r2.shape[0] == x2.shape[0] and gemm_supported(x2, w, b, r2):
                return ops().gemm_nt(x2, w, b, _EPI_ADD, r2)[0].view(*x.shape[:-1], w.shape[0])
        elif gemm_supported(x2, w, b):
            return ops().gemm_nt(x2, w, b, _EPI_NONE, None)[0].view(*x.shape[:-1], w.shape[0])
    y = F.linear(x, w, b)
    return y if residual is None else y + residual
