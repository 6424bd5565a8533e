"""Routed-expert evaluation of a MoE layer without a loop over experts and without a host sync.

The (token, slot) pairs are laid out sorted by expert, every expert's segment padded to a
multiple of 128 rows inside a buffer whose size is a static bound (``pairs + 127 * experts``
rounded up). A 128-row block then belongs to exactly one expert, so the grouped MFMA GEMMs of
``csrc/gemm_grouped.hip`` pick their weight matrix per block from a small device table and the
whole layer runs with shapes known on the host: no ``.tolist()``, no per-expert launches.

Forward:  scatter rows -> grouped ``X W_fc^T`` -> activation -> grouped ``A W_proj^T`` -> gate-weighted
gather. Backward mirrors it with the NN form for the input gradients and a segment-aware
split-M kernel for the weight gradients.

Reference behaviour: the per-expert ``nonzero`` / ``index_add_`` loop of
single-gpu/model.py:490-502.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip

BLOCK = 128


def padded_rows(npairs: int, n_experts: int, block: int = BLOCK) -> int:
    """Static row count of the padded expert-sorted buffer (expert segments padded to ``block`` rows)."""
    return (npairs + n_experts * (block - 1) + block - 1) // block * block


def expert_counts(topk_indices: torch.Tensor, n_experts: int) -> torch.Tensor:
    """Pairs per expert, int64 [E]. (``torch.bincount`` reads the maximum back to the host on the GPU;
    a scatter-add does not.)"""
    flat = topk_indices.reshape(-1)
    return torch.zeros(n_experts, dtype=torch.int64, device=flat.device).scatter_add_(
        0, flat, torch.ones_like(flat))


def plan(topk_indices: torch.Tensor, n_experts: int, block: int = BLOCK):
    """Device-side layout of the pairs (no host sync). ``block`` = rows of a GEMM row block: 128 for the
    grouped kernels of ``csrc/gemm_grouped.hip``, 256 for the grouped mode of the NT GEMM.

    -> row_of_pair [N*k] int64 (row of pair ``t*k + s`` in the padded buffer),
       block_group [rows/block] int32 (expert of each block, ``n_experts`` past the end),
       seg_start, seg_rows [E] int64 (padded segment of each expert), rows (python int),
       counts [E] int64.
    On the GPU this is the histogram / scan / placement kernel trio of ``csrc/moe_route.hip``; the
    PyTorch statement below (one stable sort) defines the same layout."""
    flat_e = topk_indices.reshape(-1)
    npairs = flat_e.numel()
    rows = padded_rows(npairs, n_experts, block)
    if use_hip(flat_e) and n_experts <= 64 and flat_e.dtype == torch.int64:
        row_of_pair, block_group, seg_start, seg_rows, counts = ops().moe_plan(flat_e.contiguous(), n_experts, rows, block)
        return row_of_pair, block_group, seg_start, seg_rows, rows, counts
    order = torch.argsort(flat_e, stable=True)
    counts = expert_counts(flat_e, n_experts)
    cstart = torch.cumsum(counts, 0) - counts
    seg_rows = (counts + (block - 1)) // block * block
    seg_end = torch.cumsum(seg_rows, 0)
    seg_start = seg_end - seg_rows
    e_sorted = flat_e[order]
    dest = seg_start[e_sorted] + (torch.arange(npairs, device=flat_e.device) - cstart[e_sorted])
    row_of_pair = torch.empty_like(order)
    row_of_pair[order] = dest
    block_first = torch.arange(rows // block, device=flat_e.device) * block
    block_group = torch.searchsorted(seg_end, block_first, right=True).to(torch.int32)
    return row_of_pair, block_group, seg_start.contiguous(), seg_rows.contiguous(), rows, counts


def plan_dense(topk_indices: torch.Tensor, n_experts: int):
    """Unpadded expert-sorted order of the (token, slot) pairs: -> (row_of_pair [N*k] int64, counts [E]
    int64). Row ``row_of_pair[t*k + s]`` of a buffer laid out this way holds pair ``(t, s)``; pairs of one
    expert are contiguous and keep their order, which is the send order of an expert-parallel
    all-to-all (experts are owned by ranks in contiguous blocks)."""
    flat_e = topk_indices.reshape(-1)
    npairs = flat_e.numel()
    if use_hip(flat_e) and n_experts <= 64 and flat_e.dtype == torch.int64 and npairs > 0:
        row_of_pair, _, _, _, counts = ops().moe_plan(flat_e.contiguous(), n_experts, npairs, 0)
        return row_of_pair, counts
    order = torch.argsort(flat_e, stable=True)
    row_of_pair = torch.empty_like(order)
    row_of_pair[order] = torch.arange(npairs, device=flat_e.device)
    return row_of_pair, expert_counts(flat_e, n_experts)


class _ScatterRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, row_of_pair, k, rows, tables):
        ctx.save_for_backward(row_of_pair)
        ctx.k = k
        return ops().moe_scatter_rows(x.contiguous(), row_of_pair, k, rows, *(tables or ()))

    @staticmethod
    def backward(ctx, g):
        (row_of_pair,) = ctx.saved_tensors
        return ops().moe_gather_combine(g.contiguous(), row_of_pair, None, ctx.k), None, None, None, None


class _GatherCombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, row_of_pair, gates, k, tables):
        yc, gc = y.contiguous(), gates.contiguous()
        ctx.save_for_backward(yc, row_of_pair, gc, *(tables or ()))
        return ops().moe_gather_combine(yc, row_of_pair, gc, k)

    @staticmethod
    def backward(ctx, dout):
        y, row_of_pair, gates, *tables = ctx.saved_tensors
        dy, dgate = ops().moe_combine_bwd(dout.contiguous(), y, row_of_pair, gates, *tables)
        return dy, None, dgate.to(gates.dtype), None, None


def _rows_hip_ok(x, row_of_pair) -> bool:
    return use_hip(x, row_of_pair) and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] % 8 == 0 \
        and row_of_pair.numel() > 0


def scatter_rows(x: torch.Tensor, row_of_pair: torch.Tensor, k: int, rows: int, tables=None) -> torch.Tensor:
    """out[row_of_pair[t*k + s]] = x[t] (other rows zero); differentiable in x. A negative row drops its pair.
    ``tables`` = (seg_start, counts, seg_rows), int64 [segments]: the buffer is a sequence of padded segments whose
    first ``counts`` rows are all written by pairs, so only the remaining rows are zero-filled instead of all."""
    if _rows_hip_ok(x, row_of_pair):
        return _ScatterRowsFn.apply(x, row_of_pair, k, rows, tables)
    out = x.new_zeros(rows + 1, x.shape[1])     # row `rows` swallows the dropped pairs
    dest = torch.where(row_of_pair < 0, torch.full_like(row_of_pair, rows), row_of_pair)
    return out.index_copy(0, dest, x.repeat_interleave(k, 0))[:rows]


def gather_combine(y: torch.Tensor, row_of_pair: torch.Tensor, gates: torch.Tensor, k: int, tables=None) -> torch.Tensor:
    """out[t] = sum_s gates[t, s] * y[row_of_pair[t*k + s]]; differentiable in y and gates. A pair with a negative
    row contributes nothing. ``tables`` as in ``scatter_rows``, for the gradient buffer of ``y``."""
    if _rows_hip_ok(y, row_of_pair) and gates.dtype == torch.bfloat16:
        return _GatherCombineFn.apply(y, row_of_pair, gates, k, tables)
    y_ext = torch.cat([y, y.new_zeros(1, y.shape[1])], 0)
    src = torch.where(row_of_pair < 0, torch.full_like(row_of_pair, y.shape[0]), row_of_pair)
    picked = y_ext[src].view(-1, k, y.shape[1])
    return (picked * gates.unsqueeze(-1).to(picked.dtype)).sum(1)


class _TopkGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, bias, k):
        idx, gates = ops().moe_topk_gate(logits.contiguous(), bias, k)
        ctx.save_for_backward(idx, gates)
        ctx.n_experts = logits.shape[1]
        ctx.mark_non_differentiable(idx)
        return idx, gates

    @staticmethod
    def backward(ctx, _didx, dgates):
        idx, gates = ctx.saved_tensors
        dsel = gates * (dgates - (gates * dgates).sum(1, keepdim=True))    # softmax over the selected logits
        dlogits = torch.zeros(idx.shape[0], ctx.n_experts, dtype=dgates.dtype, device=dgates.device)
        dlogits.scatter_(1, idx, dsel)
        return dlogits, None, None


def topk_gate(router_logits: torch.Tensor, bias: torch.Tensor | None, k: int):
    """(topk_indices int64 [N, k], gates fp32 [N, k]): top-k of ``router_logits + bias`` and the softmax of
    the selected raw logits, in one kernel on the GPU (reference: single-gpu/model.py:447-472)."""
    E = router_logits.shape[1]
    if use_hip(router_logits) and router_logits.dtype == torch.float32 and E <= 64 and k <= min(8, E):
        b = None if bias is None else bias.detach().float().contiguous()
        return _TopkGateFn.apply(router_logits, b, k)
    biased = router_logits if bias is None else router_logits + bias.float()
    _, idx = torch.topk(biased, k, dim=1)
    return idx, F.softmax(torch.gather(router_logits, 1, idx), dim=1)


_PTR_TABLES: dict = {}


def _ptr_table(ws) -> torch.Tensor:
    """int64 device tensor of the weights' addresses, cached by the addresses themselves (they only
    change when an engine re-materialises its parameters)."""
    key = tuple(w.data_ptr() for w in ws)
    t = _PTR_TABLES.get(key)
    if t is None:
        if len(_PTR_TABLES) > 512:
            _PTR_TABLES.clear()
        t = torch.tensor(key, dtype=torch.int64).pin_memory().to(ws[0].device, non_blocking=True)
        _PTR_TABLES[key] = t
    return t


def _act_fwd(h, act):
    if act == "swiglu":
        return ops().swiglu_fwd(h)
    if act == "gelu":
        return ops().gelu_fwd(h)
    from .activation import get_activation     # the other elementwise activations of the reference's table
    with torch.no_grad():
        return get_activation(act)(h)


def _act_bwd(da, h, act):
    if act == "swiglu":
        return ops().swiglu_bwd(da, h)
    if act == "gelu":
        return ops().gelu_bwd(da, h)
    from .activation import get_activation
    with torch.enable_grad():
        hh = h.detach().requires_grad_()
        a = get_activation(act)(hh)
    return torch.autograd.grad(a, hh, da)[0]


def _direct_grad(p):
    """The flat-buffer gradient view of ``p`` if its engine asked for direct accumulation, else None."""
    g = None
    if getattr(p, "_dpa_direct", False):
        g = getattr(p, "_dpa_grad32", None)
        if g is None:
            g = p.grad
    if g is not None and g.is_contiguous() and g.dtype in (torch.bfloat16, torch.float32) \
            and g.data_ptr() % 16 == 0 and g.numel() % 8 == 0:
        return g
    return None


_WGRAD3_SET = False


def _grouped_wgrad(dy, x, seg_start, seg_rows):
    """fp32 [E, N, K] per-expert weight gradients: from 96 K rows on the grouped mode of the dense 256 x 256 kernel
    (equal row chunks across the segments, atomically combined), below that the 128 x 128 grouped kernel.
    DPA_MOE_WGRAD3=0 keeps the latter everywhere (A/B runs)."""
    global _WGRAD3_SET
    if not _WGRAD3_SET:
        _WGRAD3_SET = True
        if os.environ.get("DPA_MOE_WGRAD3", "1") == "0" and hasattr(ops(), "grouped_wgrad_use_wgrad3"):
            ops().grouped_wgrad_use_wgrad3(False)
    if os.environ.get("DPA_DEBUG_SEG") == "1":     # debugging aid (host read): the segment table of this call
        print("grouped_wgrad", tuple(dy.shape), tuple(x.shape), dy.stride(), x.stride(), seg_start.tolist(), seg_rows.tolist(),
              flush=True)
    return ops().grouped_wgrad(dy, x, seg_start, seg_rows)


def _wgrads(dy, x, seg_start, seg_rows, params, needs):
    """Per-expert weight gradients of ``params`` from (dy, x): computed and emitted now, or, for parameters of an
    expert-parallel engine while its deferral window is open (``ops/linear.py::maybe_defer_wgrad``), queued until
    the exchange of the experts' input gradient is on the wire (they accumulate straight into the flat buffers,
    so autograd gets None either way)."""
    from .linear import maybe_defer_wgrad
    targets = [_direct_grad(p) if needs[i] else None for i, p in enumerate(params)]
    if all(needs) and all(t is not None for t in targets) and len({t.dtype for t in targets}) == 1:
        if maybe_defer_wgrad(list(params),
                             lambda: _emit_wgrads(_grouped_wgrad(dy, x, seg_start, seg_rows), params, needs)):
            return [None] * len(params)
    return _emit_wgrads(_grouped_wgrad(dy, x, seg_start, seg_rows), params, needs)


def _emit_wgrads(ws32, params, needs):
    """fp32 [E, N, K] gradients -> per-parameter results: accumulated straight into the engine's
    flat gradient buffer when the parameters ask for it (``_dpa_direct``; one launch for all
    experts), else bf16 tensors returned to autograd."""
    targets = [_direct_grad(p) if needs[i] else None for i, p in enumerate(params)]
    if all(t is not None for t in targets) and len({t.dtype for t in targets}) == 1:
        ops().ws_to_many(ws32, _ptr_table(targets), True, targets[0].dtype == torch.bfloat16)
        return [None] * len(params)
    out = [None] * len(params)
    lazy = None
    for i, p in enumerate(params):
        if not needs[i]:
            continue
        if targets[i] is not None:
            ops().ws_to(ws32[i], targets[i], True)
        else:
            if lazy is None:
                lazy = torch.empty(ws32.shape, dtype=torch.bfloat16, device=ws32.device)
                ops().ws_to(ws32, lazy, False)
            out[i] = lazy[i]
    return out


def _keep_rows(y, p_drop: float, keep_pairs, row_of_pair):
    """Scaled keep mask (0 or 1 / (1 - p), y's dtype) for dropout on the experts' outputs in the padded expert-sorted
    layout: drawn from torch's GPU generator (so manual_seed, recomputation and hipGraph replay treat it like any
    other random op), or placed from a given ``keep_pairs [N, k, C]`` (tests). None without dropout."""
    if p_drop <= 0.0:
        return None
    if keep_pairs is None:
        keep = torch.rand_like(y, dtype=torch.float32) >= p_drop
    else:
        keep = torch.zeros(y.shape, dtype=torch.bool, device=y.device)
        keep[row_of_pair.reshape(-1).long()] = keep_pairs.reshape(-1, y.shape[1]).to(torch.bool)
    return keep.to(y.dtype) * (1.0 / (1.0 - p_drop))


class _GroupedExpertsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_flat, gates, topk_indices, act, layout, p_drop, keep_pairs, *weights):
        E = len(weights) // 2
        fc, proj = weights[:E], weights[E:]
        k = topk_indices.shape[1]
        row_of_pair, block_group, seg_start, seg_rows, rows, counts = (layout or plan(topk_indices, E))[:6]
        fc_ptrs, proj_ptrs = _ptr_table(fc), _ptr_table(proj)
        # (with the layout's tables only the pair-less rows of the padded buffer are zeroed, not all of it)
        xs = ops().moe_scatter_rows(x_flat, row_of_pair, k, rows, seg_start, counts, seg_rows)
        h = ops().grouped_gemm(xs, fc_ptrs, fc[0].shape[0], fc[0].shape[1], False, block_group)
        a = _act_fwd(h, act)
        y = ops().grouped_gemm(a, proj_ptrs, proj[0].shape[0], proj[0].shape[1], False, block_group)
        keep = _keep_rows(y, p_drop, keep_pairs, row_of_pair)
        if keep is not None:
            y = y * keep            # dropout sits on every expert's output, before the gate (reference: Expert -> MLP)
        gates_c = gates.contiguous()
        out = ops().moe_gather_combine(y, row_of_pair, gates_c, k)
        ctx.save_for_backward(xs, h, y, gates_c, row_of_pair, block_group, seg_start, seg_rows, keep, counts)
        ctx.params = weights
        ctx.act = act
        ctx.k = k
        return out

    @staticmethod
    def backward(ctx, dout):
        xs, h, y, gates, row_of_pair, block_group, seg_start, seg_rows, keep, counts = ctx.saved_tensors
        weights = ctx.params
        E = len(weights) // 2
        fc, proj = weights[:E], weights[E:]
        # Addresses are looked up again: a sharding engine may have re-materialised the weights.
        fc_ptrs, proj_ptrs = _ptr_table(fc), _ptr_table(proj)
        needs = ctx.needs_input_grad[7:]
        dy, dgate = ops().moe_combine_bwd(dout.contiguous(), y, row_of_pair, gates, seg_start, counts, seg_rows)
        if keep is not None:
            dy = dy * keep
        a = _act_fwd(h, ctx.act)
        g_proj = [None] * E
        if any(needs[E:]):
            g_proj = _wgrads(dy, a, seg_start, seg_rows, proj, needs[E:])
        del a
        da = ops().grouped_gemm(dy, proj_ptrs, proj[0].shape[0], proj[0].shape[1], True, block_group)
        dh = _act_bwd(da, h, ctx.act)
        del da
        g_fc = [None] * E
        if any(needs[:E]):
            g_fc = _wgrads(dh, xs, seg_start, seg_rows, fc, needs[:E])
        dx = None
        if ctx.needs_input_grad[0]:
            dxs = ops().grouped_gemm(dh, fc_ptrs, fc[0].shape[0], fc[0].shape[1], True, block_group)
            dx = ops().moe_gather_combine(dxs, row_of_pair, None, ctx.k)
        dg = dgate.to(gates.dtype) if ctx.needs_input_grad[1] else None
        return (dx, dg, None, None, None, None, None, *g_fc, *g_proj)


_EPI_GELU, _EPI_MUL, _EPI_SWIGLU, _EPI_DSWIGLU = 1, 2, 4, 5   # epilogues of the NT GEMM (csrc/kernels.h)
NT_BLOCK = 256   # row-block size of the grouped mode of the NT GEMM (its output tile has 256 rows)


def _nt_grouped_ok(x_flat, fc, proj, act: str, k: int = 8) -> bool:
    """GELU or SwiGLU experts on the grouped mode of the dense NT GEMM (csrc/gemm_nt.hip): same persistent
    256 x 256 kernel as the dense layers, the weight matrix chosen per 256-row block, GELU / GELU' and the
    dGELU multiply, or SwiGLU and its backward, in its epilogues. Needs the 256-row layout."""
    if act not in ("gelu", "swiglu") or os.environ.get("DPA_MOE_NT", "1") == "0":
        return False
    if act == "swiglu" and os.environ.get("DPA_MOE_NT_SWIGLU", "1") == "0":     # A/B switch
        return False
    if not (use_hip(x_flat) and x_flat.dtype == torch.bfloat16 and hasattr(ops(), "gemm_nt_grouped")):
        return False
    U, C = fc[0].shape
    if act == "swiglu":
        if U % 2:
            return False
        U //= 2
    return C % 64 == 0 and C >= 128 and U % 64 == 0 and U >= 128 and tuple(proj[0].shape) == (C, U)


class _GroupedExpertsNtFn(torch.autograd.Function):
    """The routed experts of a layer as four grouped NT GEMMs and two grouped weight gradients."""

    @staticmethod
    def forward(ctx, x_flat, gates, topk_indices, layout, act, p_drop, keep_pairs, *weights):
        E = len(weights) // 2
        fc, proj = weights[:E], weights[E:]
        k = topk_indices.shape[1]
        row_of_pair, block_group, seg_start, seg_rows, rows, counts = (layout or plan(topk_indices, E, NT_BLOCK))[:6]
        assert block_group.numel() * NT_BLOCK == rows, "the grouped NT GEMM needs the 256-row layout"
        xs = ops().moe_scatter_rows(x_flat, row_of_pair, k, rows, seg_start, counts, seg_rows)
        C = fc[0].shape[1]
        U = proj[0].shape[1]
        # GELU: a = gelu(h), side = gelu'(h); SwiGLU: a = silu(gate) * up, side = [gate | up]
        a, side = ops().gemm_nt_grouped(xs, _ptr_table(fc), U, C, _EPI_SWIGLU if act == "swiglu" else _EPI_GELU, None,
                                        block_group)
        y = ops().gemm_nt_grouped(a, _ptr_table(proj), C, U, 0, None, block_group)[0]
        keep = _keep_rows(y, p_drop, keep_pairs, row_of_pair)
        if keep is not None:
            y = y * keep            # dropout sits on every expert's output, before the gate (reference: Expert -> MLP)
        gates_c = gates.contiguous()
        out = ops().moe_gather_combine(y, row_of_pair, gates_c, k)
        ctx.save_for_backward(xs, a, side, y, gates_c, row_of_pair, block_group, seg_start, seg_rows, keep, counts)
        ctx.params = weights
        ctx.k = k
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dout):
        from .linear import transposed_weight
        xs, a, side, y, gates, row_of_pair, block_group, seg_start, seg_rows, keep, counts = ctx.saved_tensors
        weights = ctx.params
        E = len(weights) // 2
        fc, proj = weights[:E], weights[E:]
        W, C = fc[0].shape                 # W = U, or 2U with SwiGLU experts
        U = proj[0].shape[1]
        needs = ctx.needs_input_grad[7:]
        dy, dgate = ops().moe_combine_bwd(dout.contiguous(), y, row_of_pair, gates, seg_start, counts, seg_rows)
        if keep is not None:
            dy = dy * keep
        g_proj = [None] * E
        if any(needs[E:]):
            g_proj = _wgrads(dy, a, seg_start, seg_rows, proj, needs[E:])
        # dh = (dy W_proj) * gelu'(h), or the SwiGLU backward of it: NT on the per-step cached transposes, the
        # activation's derivative in the epilogue
        proj_t = [transposed_weight(w) for w in proj]          # [U, C] each
        dh = ops().gemm_nt_grouped(dy, _ptr_table(proj_t), U, C, _EPI_DSWIGLU if ctx.act == "swiglu" else _EPI_MUL, side,
                                   block_group)[0]
        g_fc = [None] * E
        if any(needs[:E]):
            g_fc = _wgrads(dh, xs, seg_start, seg_rows, fc, needs[:E])
        dx = None
        if ctx.needs_input_grad[0]:
            fc_t = [transposed_weight(w) for w in fc]          # [C, W] each
            dxs = ops().gemm_nt_grouped(dh, _ptr_table(fc_t), C, W, 0, None, block_group)[0]
            dx = ops().moe_gather_combine(dxs, row_of_pair, None, ctx.k)
        dg = dgate.to(gates.dtype) if ctx.needs_input_grad[1] else None
        return (dx, dg, None, None, None, None, None, *g_fc, *g_proj)


def grouped_layout_block(x_flat, fc_weights, proj_weights, act: str, k: int = 8) -> int:
    """Row-block size of the padded layout that ``grouped_experts`` will want for these operands (``k`` =
    experts per token)."""
    return NT_BLOCK if _nt_grouped_ok(x_flat, fc_weights, proj_weights, act, k) else BLOCK


def grouped_supported(x_flat, fc_weights, proj_weights, act: str) -> bool:
    if act == "glu" or not use_hip(x_flat) or x_flat.dtype != torch.bfloat16:     # glu halves the width
        return False
    if os.environ.get("DPA_MOE_GROUPED", "1") == "0":   # A/B switch: per-expert library GEMMs
        return False
    C = x_flat.shape[1]
    U = proj_weights[0].shape[1]
    ok = C % 128 == 0 and U % 128 == 0 and x_flat.is_contiguous()
    for w in (*fc_weights, *proj_weights):
        ok = ok and w.dtype == torch.bfloat16 and w.is_contiguous() and w.data_ptr() % 16 == 0
    ok = ok and all(w.shape == fc_weights[0].shape for w in fc_weights)
    ok = ok and all(w.shape == proj_weights[0].shape for w in proj_weights)
    return ok


def grouped_experts(x_flat, topk_indices, topk_gates, fc_weights, proj_weights, act: str, layout=None,
                    p_drop: float = 0.0, keep_pairs=None) -> torch.Tensor:
    """sum_s gate[t, s] * drop(expert_{idx[t, s]}(x[t])) for bias-free two-layer experts
    (``act`` = "swiglu" or an elementwise activation of the reference's table) on the grouped HIP kernels. ``layout`` = a ``plan()`` result for
    ``topk_indices`` if the caller already has one. ``p_drop`` > 0: dropout on every expert's output (mask drawn on
    the device: no host work, so the layer stays capturable); ``keep_pairs [N, k, C]`` pins the mask (tests)."""
    assert len(fc_weights) == len(proj_weights)
    p_drop = float(p_drop)
    if _nt_grouped_ok(x_flat, fc_weights, proj_weights, act, topk_indices.shape[1]) and \
            (layout is None or layout[1].numel() * NT_BLOCK == layout[4]):
        return _GroupedExpertsNtFn.apply(x_flat, topk_gates, topk_indices, layout, act, p_drop, keep_pairs,
                                         *fc_weights, *proj_weights)
    if layout is not None and layout[1].numel() * BLOCK != layout[4]:
        layout = None      # made for the other kernels' block size
    return _GroupedExpertsFn.apply(x_flat, topk_gates, topk_indices, act, layout, p_drop, keep_pairs,
                                   *fc_weights, *proj_weights)


def grouped_experts_eager(x_flat, topk_indices, topk_gates, fc_weights, proj_weights, act: str, p_drop: float = 0.0,
                          keep_pairs=None) -> torch.Tensor:
    """Plain PyTorch statement of ``grouped_experts`` (dense over experts; test oracle)."""
    out = torch.zeros_like(x_flat)
    keep_scale = None
    if p_drop > 0.0:
        keep_scale = keep_pairs.to(x_flat.dtype) / (1.0 - p_drop)         # [N, k, C]
    for e, (wf, wp) in enumerate(zip(fc_weights, proj_weights)):
        h = F.linear(x_flat, wf)
        if act == "swiglu":
            u, v = h.chunk(2, dim=-1)
            a = F.silu(u) * v
        elif act == "gelu":
            a = F.gelu(h)
        else:
            from .activation import _KINDS
            a = _KINDS[act][1](h)
        y = F.linear(a, wp)
        if keep_scale is None:
            w = (topk_gates * (topk_indices == e)).sum(1, keepdim=True)
            out = out + w.to(y.dtype) * y
        else:
            w = topk_gates * (topk_indices == e)                          # [N, k]: the slot that chose expert e
            out = out + (w.unsqueeze(-1).to(y.dtype) * keep_scale).sum(1) * y
    return out
