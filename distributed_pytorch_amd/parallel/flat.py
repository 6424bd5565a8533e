"""Flat parameter / gradient storage shared by every training engine.

The parameters of a model are packed into a few large contiguous *groups* (one per gradient
bucket or FSDP unit). Inside a group the tensors that receive weight decay come first, then
the ones that do not, so the fused AdamW kernel needs two ``(end, weight_decay)`` segments per
group and nothing per tensor. Every tensor starts on a 128-byte boundary and every group is
padded to a multiple of ``8 * world_size`` elements, so a group can be cut into equal,
vector-aligned shards for reduce-scatter / all-gather without repacking.

* ``FlatGroup.data``  : compute-dtype buffer; each ``Parameter.data`` is a view into it.
* ``FlatGroup.grad``  : gradient buffer of the same dtype; each ``Parameter.grad`` is a view,
  so autograd accumulates straight into the buffer that the collectives send.
* ``FlatGroup.grad32``: with ``grad_dtype=torch.float32`` on bf16 parameters, the fp32 buffer that
  accumulates over micro-steps, is reduced across ranks and feeds the optimizer (the reference
  accumulates and all-reduces fp32 gradients: fp32 parameters under autocast, reference:
  single-gpu/train.py:338-345). The weight-gradient kernels add their fp32 partial sums straight
  into it (``Parameter._dpa_grad32``); what autograd accumulates into the bf16 ``grad`` views
  (embedding rows, LayerNorm, ...) is folded in by ``fold()`` before a collective or the step.
  ``FlatGroup.rgrad`` names whichever buffer is the one to reduce and to step on.
* ``FlatAdamW``       : fp32 master / exp_avg / exp_avg_sq for an element range of a group
  (the whole group, or this rank's shard) + the fused step.

The decay rule is the reference's: tensors with >= 2 dims decay, others do not
(reference: single-gpu/model.py:619-629) — decided on the ORIGINAL shapes, so it stays correct
when a group is sharded (the reference's FSDP path decides on flattened 1-D views).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Sequence

import torch
import torch.nn as nn

from .. import ops

ALIGN = 64  # elements; 128 B for bf16, 256 B for fp32


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass
class Slot:
    name: str
    param: nn.Parameter
    offset: int
    numel: int
    shape: torch.Size
    decay: bool


class FlatGroup:
    """One contiguous group of parameters (a bucket / FSDP unit)."""

    resident = True   # the parameters stay at their addresses for the whole run (sharded units: released and re-gathered)

    def __init__(self, named_params: Sequence[tuple[str, nn.Parameter]], dtype: torch.dtype, device,
                 pad_multiple: int = 8, weight_decay: float = 0.0, name: str = "", grad_dtype: torch.dtype | None = None):
        self.name = name
        self.dtype = dtype
        self.wide_grad = grad_dtype == torch.float32 and dtype != torch.float32
        self.grad32: torch.Tensor | None = None
        self.device = torch.device(device)
        self.weight_decay = weight_decay
        decay = [(n, p) for n, p in named_params if p.dim() >= 2]
        nodecay = [(n, p) for n, p in named_params if p.dim() < 2]
        self.slots: list[Slot] = []
        off = 0
        for is_decay, lst in ((True, decay), (False, nodecay)):
            for n, p in lst:
                self.slots.append(Slot(n, p, off, p.numel(), p.shape, is_decay))
                off = _round_up(off + p.numel(), ALIGN)
            if is_decay:
                self.decay_end = off
        self.numel = _round_up(max(off, pad_multiple), pad_multiple)
        self.data = torch.zeros(self.numel, dtype=dtype, device=self.device)
        self.grad: torch.Tensor | None = None
        # fp32 snapshot of the initial weights (taken before any down-cast) for the master copy
        self._init_fp32 = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for s in self.slots:
                src = s.param.data
                self._init_fp32[s.offset:s.offset + s.numel].copy_(src.reshape(-1).float())
                view = self.data[s.offset:s.offset + s.numel].view(s.shape)
                view.copy_(src)
                s.param.data = view
                s.param._dpa_resident = self.resident
        self.attach_grads()

    # -- gradient buffer --------------------------------------------------------------------
    def attach_grads(self):
        if self.grad is None:
            self.grad = torch.zeros(self.numel, dtype=self.dtype, device=self.device)
        if self.wide_grad and self.grad32 is None:
            self.grad32 = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self._attach_views()

    def _attach_views(self):
        for s in self.slots:
            s.param.grad = self.grad[s.offset:s.offset + s.numel].view(s.shape)
            if self.grad32 is not None:
                s.param._dpa_grad32 = self.grad32[s.offset:s.offset + s.numel].view(s.shape)

    @property
    def rgrad(self) -> torch.Tensor:
        """The gradient buffer that is reduced across ranks and handed to the optimizer."""
        return self.grad32 if self.grad32 is not None else self.grad

    def fold(self):
        """fp32 gradients: move what autograd accumulated in the bf16 views into the fp32 buffer."""
        if self.grad32 is not None:
            self.grad32.add_(self.grad)
            self.grad.zero_()

    def zero_grad(self):
        if self.grad32 is not None:
            self.grad32.zero_()
        if self.grad is not None:
            self.grad.zero_()
            for s in self.slots:  # re-attach in case somebody replaced .grad
                if s.param.grad is None or s.param.grad.data_ptr() != self.grad.data_ptr() + s.offset * self.grad.element_size():
                    s.param.grad = self.grad[s.offset:s.offset + s.numel].view(s.shape)

    def take_init_fp32(self, lo: int, hi: int) -> torch.Tensor:
        return self._init_fp32[lo:hi].clone()

    def drop_init(self):
        self._init_fp32 = None

    def segments(self, lo: int = 0, hi: int | None = None) -> list[tuple[int, float]]:
        """(end, weight_decay) segments of the element range [lo, hi), relative to lo."""
        hi = self.numel if hi is None else hi
        d = min(max(self.decay_end - lo, 0), hi - lo)
        segs = []
        if d > 0:
            segs.append((d, self.weight_decay))
        if hi - lo > d or not segs:
            segs.append((hi - lo, 0.0))
        return segs


class FlatAdamW:
    """fp32 AdamW state for the element range [lo, hi) of a FlatGroup."""

    def __init__(self, group: FlatGroup, lo: int = 0, hi: int | None = None, betas=(0.9, 0.999), eps: float = 1e-8):
        self.group = group
        self.lo = lo
        self.hi = group.numel if hi is None else hi
        n = self.hi - self.lo
        assert n % 8 == 0 and lo % 8 == 0, "optimizer ranges must be multiples of 8 elements"
        self.master = group.take_init_fp32(self.lo, self.hi)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=group.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=group.device)
        self.betas = betas
        self.eps = eps
        self.segs = group.segments(self.lo, self.hi)

    def step(self, grad: torch.Tensor, param_out: torch.Tensor | None, lr: float, step: int,
             norm2: torch.Tensor | None = None, max_norm: float = 0.0, grad_mul: float = 1.0,
             dyn: torch.Tensor | None = None):
        ops.adamw_step(self.master, self.exp_avg, self.exp_avg_sq, grad, param_out, self.segs, lr, self.betas[0],
                       self.betas[1], self.eps, step, norm2, max_norm, grad_mul, dyn)

    def state_dict(self):
        return {"master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lo": self.lo, "hi": self.hi}

    def load_state_dict(self, sd):
        assert sd["lo"] == self.lo and sd["hi"] == self.hi, "optimizer shard layout changed"
        self.master.copy_(sd["master"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def unique_named_parameters(module: nn.Module) -> list[tuple[str, nn.Parameter]]:
    """named_parameters() with tied tensors listed once (first name wins), trainable only."""
    seen, out = set(), []
    for n, p in module.named_parameters(remove_duplicate=False):
        if p.requires_grad and id(p) not in seen:
            seen.add(id(p))
            out.append((n, p))
    return out


def plan_groups(model: nn.Module, bucket_elems: int, per_block: bool = False
                ) -> list[tuple[str, list[tuple[str, nn.Parameter]]]]:
    """Split the model's parameters into groups in registration order.

    The embedding side (tied token table, position table) forms the first group, consecutive
    transformer blocks are merged until a group reaches ``bucket_elems`` (or one block per
    group with ``per_block``), and the final LayerNorm joins the last block group. Backward
    produces gradients in the reverse of this order, so groups become ready last-to-first.
    """
    named = unique_named_parameters(model)
    blocks = getattr(getattr(model, "transformer", None), "h", None)
    if blocks is None:
        # generic module: greedy split in registration order
        groups, cur, cur_n = [], [], 0
        for n, p in named:
            cur.append((n, p))
            cur_n += p.numel()
            if cur_n >= bucket_elems:
                groups.append((f"g{len(groups)}", cur))
                cur, cur_n = [], 0
        if cur:
            groups.append((f"g{len(groups)}", cur))
        return groups
    block_ids = {}
    for i, b in enumerate(blocks):
        for p in b.parameters():
            block_ids[id(p)] = i
    root = [(n, p) for n, p in named if id(p) not in block_ids and not n.startswith("transformer.ln_f")]
    tail = [(n, p) for n, p in named if n.startswith("transformer.ln_f")]
    groups = [("embed", root)] if root else []
    cur, cur_n, first = [], 0, 0
    nb = len(blocks)
    for i in range(nb):
        bp = [(n, p) for n, p in named if block_ids.get(id(p)) == i]
        cur += bp
        cur_n += sum(p.numel() for _, p in bp)
        if per_block or cur_n >= bucket_elems or i == nb - 1:
            if i == nb - 1:
                cur += tail
                tail = []
            groups.append((f"blocks{first}-{i}", cur))
            cur, cur_n, first = [], 0, i + 1
    if tail:
        groups.append(("tail", tail))
    return groups


def grad_norm_sq(groups: Iterable[FlatGroup], out: torch.Tensor, ranges=None) -> torch.Tensor:
    """out = sum over groups of ||grad[range]||^2 (device scalar, no host sync)."""
    first = True
    for i, g in enumerate(groups):
        lo, hi = (0, g.numel) if ranges is None else ranges[i]
        ops.sumsq(g.rgrad[lo:hi], out, accumulate=not first)
        first = False
    if first:
        out.zero_()
    return out
