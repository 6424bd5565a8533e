"""Cross-entropy and the fused lm_head + cross-entropy.

``cross_entropy(logits, targets)``: mean NLL over targets != ignore_index. On the GPU the
softmax statistics, the loss and the gradient are produced by one kernel that overwrites the
logits with d(loss)/d(logits), so no fp32 probability tensor ever exists.

``linear_cross_entropy(x, w, targets)``: loss of ``lm_head`` + cross-entropy without keeping
the ``[B*T, V]`` logits: rows are processed in chunks; for each chunk the logits GEMM (the own NT
MFMA kernel), the cross-entropy kernel (one read, one in-place write of the chunk) and (when
training) the two gradient GEMMs ``dx = dlogits @ w`` and ``dw += dlogits^T @ x`` run back to
back on one reused bf16 buffer, so only ``[chunk, V]`` ever exists. All three products are the own kernels
(csrc/gemm_nt.hip; csrc/gemm_wgrad.hip with fp32 sums across the chunks). The cross-entropy kernel writes the
gradient already multiplied by 1 / valid targets and by the factor the caller announces for the loss
(``expected_grad``), so the backward pass has nothing left to do but check that factor. The chunk is far larger than any cache of
the part (65536 x 50304 bf16 = 6.6 GB of the 288 GB): every pass over it is an HBM pass, and the
chunk size is chosen for GEMM efficiency, not residency — measured per piece in
benchmarks/lmhead_bench.py (2048-row chunks that would fit the 256 MB last-level cache cost 25 ms
per 65536 rows against 15.4 ms with 16384-row chunks: the dx product then has too few output
tiles and the fp32 dW accumulator is re-read 32 times; 65536-row chunks gain another 1 % of the
step over 16384).

``vocab_parallel_cross_entropy``: the same loss with the vocabulary sharded over a process
group (all-reduce of max / sum-exp / target logit between the two kernel halves).

``vocab_parallel_linear_cross_entropy``: lm_head + loss of a data-parallel job with the vocabulary
split over the ranks (``--vocab_parallel``): the hidden rows of all ranks are gathered, every rank
forms the logits of all rows against its slice of the head weight (own GEMM kernels), the softmax
statistics of a chunk travel in one asynchronous all-gather under the next chunk's logits GEMM, and the
input gradient returns through a reduce-scatter. Per-rank FLOPs equal the replicated head's; each rank
produces the head's weight gradient for its own vocabulary slice only, complete over the global batch.
It costs more wire than it saves above V / W tokens per rank (see the function's docstring).

Reference behaviour: ``F.cross_entropy(logits.view(-1, V), targets.view(-1), ignore_index=-1)``
on the full fp32/amp logits (reference: single-gpu/model.py:687-692).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .._ext import ops, use_hip

IGNORE_INDEX = -1


def _hip_ok(logits: torch.Tensor, targets: torch.Tensor) -> bool:
    return (use_hip(logits, targets) and logits.dtype in (torch.bfloat16, torch.float32)
            and logits.shape[-1] % 8 == 0)


def _grad_scale(targets: torch.Tensor, ignore_index: int) -> torch.Tensor:
    n_valid = (targets != ignore_index).sum().clamp_min(1)
    return (1.0 / n_valid.float()).reshape(1)


class _XentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        need_grad = logits.requires_grad
        work = logits.contiguous()
        if need_grad and work.data_ptr() == logits.data_ptr():
            work = work.clone()  # the kernel overwrites its input; never clobber the caller's logits
        gs = _grad_scale(targets, ignore_index)
        loss_rows = ops().xent_fused(work, targets.contiguous(), gs, ignore_index, need_grad)
        if need_grad:
            ctx.save_for_backward(work)
        return loss_rows.sum() * gs[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g.to(dlogits.dtype), None, None


def cross_entropy(logits: torch.Tensor, targets: torch.Tensor, ignore_index: int = IGNORE_INDEX) -> torch.Tensor:
    logits2 = logits.reshape(-1, logits.shape[-1])
    targets1 = targets.reshape(-1)
    if _hip_ok(logits2, targets1):
        return _XentFn.apply(logits2, targets1, ignore_index)
    return F.cross_entropy(logits2.float(), targets1, ignore_index=ignore_index)


def _library_head() -> bool:
    import os
    return os.environ.get("DPA_LIBRARY_GEMM", "0") == "1" or os.environ.get("DPA_LOSS_OWN_GEMM", "1") == "0"


def _head_plan(x: torch.Tensor, w: torch.Tensor):
    """Which of the three products of the head run on the own kernels: (logits, dx, dw)."""
    if x.dtype != torch.bfloat16 or _library_head() or not hasattr(ops(), "gemm_nt_into"):
        return False, False, False
    V, C = w.shape
    probe = x[: min(x.shape[0], 16384)]
    own_logits = bool(ops().gemm_nt_supported(probe, w, None, None))
    # dx = dlogits [rows, V] @ w through the NT kernel on w^T [C, V]: contraction over the vocabulary
    own_dx = V % 64 == 0 and V >= 128 and C % 16 == 0 and w.is_contiguous()
    # dw += dlogits^T x on the 256 x 256 weight-gradient kernel, fp32 sums kept in the kernel's own walk over the tiles
    own_dw = V % 128 == 0 and C % 256 == 0
    return own_logits, own_dx, own_dw


class _LinearXentFn(torch.autograd.Function):
    """Row chunks of: logits GEMM -> cross-entropy kernel (one read, one in-place write of the chunk: loss, and
    d loss / d logits already multiplied by 1 / valid targets and by the factor the caller announced for the loss)
    -> dx GEMM -> dw GEMM (fp32 sums across the chunks). All three products are the own kernels; backward only
    checks that the incoming gradient is the announced factor (``scale_unless``: no pass over dx / dw when it is)."""

    @staticmethod
    def forward(ctx, x, w, targets, ignore_index, chunk_rows, expected_grad):
        N, C = x.shape
        V = w.shape[0]
        need_dx = ctx.needs_input_grad[0]
        need_dw = ctx.needs_input_grad[1]
        train = need_dx or need_dw
        gs = _grad_scale(targets, ignore_index)
        gs_grad = gs * expected_grad if expected_grad != 1.0 else gs
        loss_acc = torch.zeros((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if need_dx else None
        own_logits, own_dx, own_dw = _head_plan(x, w)
        wt = None
        if need_dx and own_dx:
            from .linear import transposed_weight
            wt = transposed_weight(w)
        bf16 = x.dtype == torch.bfloat16
        ws = torch.zeros(V, C, dtype=torch.float32, device=x.device) if (need_dw and bf16) else None
        dw = None
        logits = torch.empty(min(chunk_rows, N), V, dtype=x.dtype, device=x.device)
        for r0 in range(0, N, chunk_rows):
            r1 = min(N, r0 + chunk_rows)
            rows = r1 - r0
            xc = x[r0:r1]
            lg = logits[:rows]
            if own_logits and ops().gemm_nt_supported(xc, w, None, None):
                ops().gemm_nt_into(xc, w, lg)        # walks row ranges of the 6.6 GB chunk by itself
            else:
                torch.mm(xc, w.t(), out=lg)
            loss_rows = ops().xent_fused(lg, targets[r0:r1], gs_grad, ignore_index, train)
            loss_acc += loss_rows.sum()
            if need_dx:
                if wt is not None and ops().gemm_nt_supported(lg, wt, None, None):
                    ops().gemm_nt_into(lg, wt, dx[r0:r1])
                else:
                    torch.mm(lg, w, out=dx[r0:r1])
            if need_dw:
                if ws is not None:
                    m64 = rows // 64 * 64 if own_dw else 0
                    if m64 >= 1024:
                        ops().wgrad2_accum(lg[:m64], xc[:m64], ws)
                    else:
                        m64 = 0
                    if m64 < rows:   # a ragged tail of fewer than 64 rows (or a tiny chunk)
                        torch.addmm(ws, lg[m64:].t(), xc[m64:], out_dtype=torch.float32, out=ws)
                elif dw is None:
                    dw = torch.mm(lg.t(), xc)
                else:
                    dw.addmm_(lg.t(), xc)
        if ws is not None:
            dw = ws.to(x.dtype)
        ctx.save_for_backward(dx, dw)
        # the factor the stored gradients currently carry (device scalar: a second backward over a retained graph
        # rescales from what the first one left, not from the announced factor again)
        ctx.applied = torch.full((1,), float(expected_grad), dtype=torch.float32, device=x.device)
        return loss_acc * gs[0]     # the loss carries 1 / valid targets only; expected_grad went into the gradients

    @staticmethod
    def backward(ctx, g):
        dx, dw = ctx.saved_tensors
        g32 = g.detach().float().reshape(1)
        for t in (dx, dw):
            if t is None:
                continue
            if t.is_cuda and t.numel() % 8 == 0 and t.is_contiguous() and t.data_ptr() % 16 == 0:
                ops().scale_unless(t, g32, ctx.applied)
            else:
                t.mul_((g32[0] / ctx.applied[0]).to(t.dtype))
        ctx.applied.copy_(g32)
        return dx, dw, None, None, None, None


def linear_cross_entropy(x: torch.Tensor, w: torch.Tensor, targets: torch.Tensor,
                         ignore_index: int = IGNORE_INDEX, chunk_rows: int = 65536,
                         expected_grad: float = 1.0) -> torch.Tensor:
    """Mean cross-entropy of ``x @ w.T`` against ``targets`` (``x [N, C]``, ``w [V, C]``).

    ``expected_grad``: the gradient the caller expects to send into this loss (1 / accumulation steps when the loss
    is divided before ``backward``). The stored gradients are produced with that factor already applied; any other
    incoming gradient is still honoured (one rescaling pass)."""
    x2 = x.reshape(-1, x.shape[-1])
    t1 = targets.reshape(-1)
    if use_hip(x2, w, t1) and x2.dtype == w.dtype and x2.dtype in (torch.bfloat16, torch.float32) \
            and w.shape[0] % 8 == 0:
        return _LinearXentFn.apply(x2.contiguous(), w, t1.contiguous(), ignore_index, int(chunk_rows),
                                   float(expected_grad) if expected_grad else 1.0)
    return F.cross_entropy(F.linear(x2, w).float(), t1, ignore_index=ignore_index)


# ---------------------------------------------------------------------------------------------
def _comm():
    from ..parallel import comm   # imported late: ops must stay importable without the parallel package
    return comm


class _VocabParallelXentFn(torch.autograd.Function):
    """logits_local [N, V/W] holds columns [vocab_start, vocab_start + V/W) of the full logits."""

    @staticmethod
    def forward(ctx, logits, targets, vocab_start, group, ignore_index):
        need_grad = logits.requires_grad
        gs = _grad_scale(targets, ignore_index)
        if _hip_ok(logits, targets):
            work = logits.contiguous()
            if need_grad and work.data_ptr() == logits.data_ptr():
                work = work.clone()
            mx, sm, tg = ops().xent_stats(work, targets, vocab_start, ignore_index)
        else:
            work = logits.float()
            mx = work.max(dim=1).values
            sm = torch.exp(work - mx[:, None]).sum(dim=1)
            loc = targets - vocab_start
            inr = (loc >= 0) & (loc < work.shape[1]) & (targets != ignore_index)
            tg = torch.where(inr, work.gather(1, loc.clamp(0, work.shape[1] - 1)[:, None])[:, 0],
                             torch.zeros_like(mx))
        gmx = mx.clone()
        _comm().all_reduce(gmx, op=dist.ReduceOp.MAX, group=group)
        sm = sm * torch.exp(mx - gmx)
        packed = torch.stack([sm, tg])
        _comm().all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        lse = gmx + torch.log(packed[0])
        tgt_logit = packed[1].contiguous()
        valid = targets != ignore_index
        if _hip_ok(logits, targets):
            loss_rows = ops().xent_grad(work, targets, lse.contiguous(), tgt_logit, gs, vocab_start, ignore_index,
                                        need_grad)
            dlogits = work
        else:
            loss_rows = torch.where(valid, lse - tgt_logit, torch.zeros_like(lse))
            dlogits = None
            if need_grad:
                p = torch.exp(work - lse[:, None])
                loc = targets - vocab_start
                inr = (loc >= 0) & (loc < work.shape[1]) & valid
                rows = torch.nonzero(inr).squeeze(1)
                p[rows, loc[rows]] -= 1.0
                p = p * (valid[:, None].to(p.dtype) * gs[0])
                dlogits = p.to(logits.dtype)
        if need_grad:
            ctx.save_for_backward(dlogits)
        return loss_rows.sum() * gs[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g.to(dlogits.dtype), None, None, None, None


def vocab_parallel_cross_entropy(logits_local: torch.Tensor, targets: torch.Tensor, vocab_start: int,
                                 group=None, ignore_index: int = IGNORE_INDEX) -> torch.Tensor:
    return _VocabParallelXentFn.apply(logits_local.reshape(-1, logits_local.shape[-1]), targets.reshape(-1),
                                      int(vocab_start), group, ignore_index)


def _local_stats(work, targets, vocab_start, ignore_index):
    """Row max, sum-exp (relative to that max) and target logit of a vocabulary slice."""
    if _hip_ok(work, targets):
        return ops().xent_stats(work, targets, vocab_start, ignore_index)
    wf = work.float()
    mx = wf.max(dim=1).values
    sm = torch.exp(wf - mx[:, None]).sum(dim=1)
    loc = targets - vocab_start
    inr = (loc >= 0) & (loc < wf.shape[1]) & (targets != ignore_index)
    tg = torch.where(inr, wf.gather(1, loc.clamp(0, wf.shape[1] - 1)[:, None])[:, 0], torch.zeros_like(mx))
    return mx, sm, tg


def _local_grad(work, targets, lse, tgt_logit, gs, vocab_start, ignore_index):
    """(loss per row, d loss / d logits of the slice in work's dtype); may overwrite ``work``."""
    if _hip_ok(work, targets):
        loss_rows = ops().xent_grad(work, targets, lse.contiguous(), tgt_logit.contiguous(), gs, vocab_start,
                                    ignore_index, True)
        return loss_rows, work
    valid = targets != ignore_index
    loss_rows = torch.where(valid, lse - tgt_logit, torch.zeros_like(lse))
    p = torch.exp(work.float() - lse[:, None])
    loc = targets - vocab_start
    rows = torch.nonzero((loc >= 0) & (loc < work.shape[1]) & valid).squeeze(1)
    p[rows, loc[rows]] -= 1.0
    p = p * (valid[:, None].to(p.dtype) * gs[0])
    return loss_rows, p.to(work.dtype)


def vocab_slices(V: int, W: int):
    """[lo, hi) of the vocabulary rows each of W ranks owns. Large vocabularies are cut at multiples of 128 rows (the
    own GEMM kernels' contraction / tile granularity), so the last rank's slice is shorter; small ones evenly."""
    per = (V + W - 1) // W
    if per >= 1024:
        per = (per + 127) // 128 * 128
    return [(min(V, r * per), min(V, (r + 1) * per)) for r in range(W)]


class _VocabParallelLinearXentFn(torch.autograd.Function):
    """Per row chunk of the gathered hidden rows: logits against this rank's vocabulary slice (own NT GEMM) -> local
    (max, sum-exp, target logit) -> ONE asynchronous all-gather of the three statistics, consumed only after the NEXT
    chunk's logits GEMM has been queued (two logits buffers alternate) -> combine -> gradient kernel -> dx partial
    (own NT GEMM on the transposed slice) and dw slice (own weight-gradient kernel, fp32 sums)."""

    @staticmethod
    def forward(ctx, x, w, targets, group, ignore_index, chunk_rows):
        comm = _comm()
        W, r = comm.world_size(group), comm.get_rank(group)
        N, C = x.shape
        V = w.shape[0]
        v0, v1 = vocab_slices(V, W)[r]
        per = v1 - v0
        ws = w[v0:v1]                                        # this rank's vocabulary slice of the tied weight
        xa = torch.empty(W * N, C, dtype=x.dtype, device=x.device)
        comm.all_gather(xa.view(-1), x.contiguous().view(-1), group=group)
        ta = torch.empty(W * N, dtype=targets.dtype, device=targets.device)
        comm.all_gather(ta, targets.contiguous(), group=group)
        gs = _grad_scale(ta, ignore_index)                   # 1 / valid targets of the global batch
        dxa = torch.empty_like(xa)
        dws = torch.zeros(per, C, dtype=torch.float32, device=x.device)
        loss_acc = torch.zeros((), dtype=torch.float32, device=x.device)
        gpu = x.is_cuda and x.dtype == torch.bfloat16 and not _library_head() and per > 0
        own_logits = gpu and bool(ops().gemm_nt_supported(xa[: min(W * N, 16384)], ws, None, None))
        wst = None
        if gpu and per % 64 == 0 and per >= 128 and C % 16 == 0:
            from .linear import transposed_weight
            wst = transposed_weight(w)[:, v0:v1]             # [C, per]: rows of w^T, row stride V
        own_dw = gpu and per % 8 == 0 and C % 256 == 0
        rows_max = min(chunk_rows, W * N)
        bufs = [torch.empty(rows_max, per, dtype=x.dtype, device=x.device) for _ in range(2 if W * N > chunk_rows else 1)]
        starts = list(range(0, W * N, chunk_rows))

        def front(ci):
            """Logits of chunk ci, its local statistics, and the exchange of them (in flight on return)."""
            r0 = starts[ci]
            r1 = min(W * N, r0 + chunk_rows)
            xc, tc = xa[r0:r1], ta[r0:r1]
            lg = bufs[ci % len(bufs)][: r1 - r0]
            if per == 0:
                mx = torch.full((r1 - r0,), float("-inf"), device=x.device)
                sm = torch.zeros(r1 - r0, device=x.device)
                tg = torch.zeros(r1 - r0, device=x.device)
            else:
                if own_logits and ops().gemm_nt_supported(xc, ws, None, None):
                    ops().gemm_nt_into(xc, ws, lg)
                else:
                    torch.mm(xc, ws.t(), out=lg)
                mx, sm, tg = _local_stats(lg, tc, v0, ignore_index)
            mine = torch.stack([mx.float(), sm.float(), tg.float()]).contiguous()          # [3, rows]
            every = torch.empty(W, 3, r1 - r0, dtype=torch.float32, device=x.device)
            work = comm.all_gather(every.view(-1), mine.view(-1), group=group, async_op=True)
            return (r0, r1, xc, tc, lg, every, work)

        def back(st):
            r0, r1, xc, tc, lg, every, work = st
            work.wait()
            gmx = every[:, 0].max(dim=0).values
            sm = (every[:, 1] * torch.exp(every[:, 0] - gmx)).sum(dim=0)      # exp(-inf - finite) = 0 for empty slices
            lse = gmx + torch.log(sm)
            tgt_logit = every[:, 2].sum(dim=0)
            if per == 0:
                valid = tc != ignore_index
                return torch.where(valid, lse - tgt_logit, torch.zeros_like(lse)).sum()
            loss_rows, dlg = _local_grad(lg, tc, lse, tgt_logit, gs, v0, ignore_index)
            if wst is not None and ops().gemm_nt_supported(dlg, wst, None, None):
                ops().gemm_nt_into(dlg, wst, dxa[r0:r1])     # partial over this slice; summed by the reduce-scatter
            else:
                torch.mm(dlg, ws.to(dlg.dtype), out=dxa[r0:r1])
            rows = r1 - r0
            m64 = rows // 64 * 64 if (own_dw and rows >= 1024) else 0
            if m64:
                ops().wgrad2_accum(dlg[:m64], xc[:m64], dws)
            if m64 < rows:
                if x.is_cuda and x.dtype == torch.bfloat16:
                    torch.addmm(dws, dlg[m64:].t(), xc[m64:], out_dtype=torch.float32, out=dws)   # bf16 operands, fp32 sum
                else:
                    dws.addmm_(dlg[m64:].t().float(), xc[m64:].float())
            return loss_rows.sum()

        pending = None
        for ci in range(len(starts)):
            st = front(ci)                      # chunk ci's logits GEMM is queued before chunk ci-1's statistics are waited for
            if pending is not None:
                loss_acc += back(pending)
            pending = st
        if pending is not None:
            loss_acc += back(pending)
        if per == 0:
            dxa.zero_()
        dx = torch.empty_like(x)
        comm.reduce_scatter(dx.view(-1), dxa.view(-1), group=group)
        # The data-parallel engines average parameter gradients over the ranks and expect each rank's
        # backward to start from the gradient of ITS loss: both are W times what the global loss gives.
        dx.mul_(W)
        dw = torch.zeros_like(w)
        dw[v0:v1] = (dws * W).to(w.dtype)
        ctx.save_for_backward(dx, dw)
        return loss_acc * gs[0]

    @staticmethod
    def backward(ctx, g):
        dx, dw = ctx.saved_tensors
        return dx * g.to(dx.dtype), dw * g.to(dw.dtype), None, None, None, None


def vocab_parallel_linear_cross_entropy(x: torch.Tensor, w: torch.Tensor, targets: torch.Tensor, group=None,
                                        ignore_index: int = IGNORE_INDEX, chunk_rows: int = 65536) -> torch.Tensor:
    """Global-batch mean cross-entropy of ``x @ w.T`` with the vocabulary dimension of ``w [V, C]`` split over
    the ranks of ``group`` (every rank holds the full tied weight and uses the rows ``vocab_slices(V, W)[rank]``).

    Wire cost per rank and step against the replicated head (N = tokens per rank, bf16): the gather of the hidden rows
    and the reduce-scatter of their gradient move ``2 (W - 1) N C 2`` bytes, the statistics ``12 (W - 1) N`` bytes; the
    replicated head moves none of that and instead all-reduces the head's share of the tied gradient,
    ``2 (W - 1) / W V C 2`` bytes. The split head therefore saves wire only below ``N = V / W`` tokens per rank
    (6288 at W = 8 for GPT-2's vocabulary: a micro-batch of 6 sequences); with tied embeddings the [V, C] gradient
    of the embedding still has to be reduced, so in practice it saves none. What it does buy is memory: the logits
    chunk is W times narrower."""
    x2 = x.reshape(-1, x.shape[-1])
    return _VocabParallelLinearXentFn.apply(x2, w, targets.reshape(-1), group, ignore_index, int(chunk_rows))
