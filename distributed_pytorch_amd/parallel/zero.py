"""ZeRO stage 1 and stage 2 on flat buffers.

Stage 1 (optimizer-state sharding): gradients are all-reduced per bucket exactly as in
``DDPEngine``; each rank keeps fp32 master / exp_avg / exp_avg_sq for only 1/W of every bucket,
runs the fused AdamW on that slice and the updated low-precision weights are re-assembled with
one in-place all-gather per bucket (instead of the reference's per-owner broadcasts).

Stage 2 (gradient sharding): the per-bucket all-reduce is replaced by a reduce-scatter
straight into the owner's gradient shard, overlapped with backward the same way; the step and
the all-gather are those of stage 1. Wire traffic per step drops from 2x to 1x the gradient
bytes plus 1x the parameter bytes.

The reference's "ZeRO-1" is DDP + ``ZeroRedundancyOptimizer`` and its "ZeRO-2" only adds
``gradient_as_bucket_view=True`` (gradients are still fully all-reduced)
(reference: multi-gpu/zero1,2/kaggle-zero1.py:1062-1077, kaggle-zero2.py:1062).
"""
from __future__ import annotations

import torch

from . import comm
from .ddp import DDPEngine
from .flat import FlatGroup


class ZeroEngine(DDPEngine):
    name = "zero"

    def __init__(self, model, stage: int = 1, **kw):
        assert stage in (1, 2)
        self.stage = stage
        super().__init__(model, **kw)
        self.name = f"zero{stage}"

    def _opt_range(self, g: FlatGroup):
        n = g.numel // self.world
        return self.rank * n, (self.rank + 1) * n

    def _build_storage(self):
        super()._build_storage()
        self._gather_works = []

    def _reduce_group(self, gi: int):
        g, o = self.groups[gi], self.optims[gi]
        g.fold()
        if self.stage == 1:
            return comm.all_reduce(g.rgrad, group=self.group, async_op=True)
        # in-place reduce-scatter: the owner's slice of the bucket receives the sum
        return comm.reduce_scatter(g.rgrad[o.lo:o.hi], g.rgrad, group=self.group, async_op=True)

    # the global norm is the sum over ranks of the norms of the owned slices
    def _compute_norm(self):
        first = True
        from .. import ops
        if not self._dist:
            self._fold_grads()
        for g, o in zip(self.groups, self.optims):
            ops.sumsq(g.rgrad[o.lo:o.hi], self._norm2, accumulate=not first)
            first = False
        comm.all_reduce(self._norm2, group=self.group)
        self._have_norm = True

    def _apply_updates(self, lr: float):
        # start each bucket's all-gather right after its update, so the gathers overlap the AdamW of
        # the following buckets
        for gi in range(len(self.groups)):
            g, o = self.groups[gi], self.optims[gi]
            o.step(g.rgrad[o.lo:o.hi], g.data[o.lo:o.hi], lr, self.step_count, self._norm2, self._max_norm,
                   self._grad_mul())
            self._gather_works.append(comm.all_gather(g.data, g.data[o.lo:o.hi], group=self.group, async_op=True))

    def _post_step(self):
        from ..utils.profiling import comm_wait_range
        with comm_wait_range():       # parameter all-gathers the remaining AdamW launches did not hide
            for w in self._gather_works:
                w.wait()
        self._gather_works = []
        super()._post_step()

    def _full_master(self, gi: int) -> torch.Tensor:
        g, o = self.groups[gi], self.optims[gi]
        full = torch.empty(g.numel, dtype=torch.float32, device=g.device)
        comm.all_gather(full, o.master, group=self.group)
        return full

    def _refresh_params_from_master(self):
        for g, o in zip(self.groups, self.optims):
            g.data[o.lo:o.hi].copy_(o.master)
            comm.all_gather(g.data, g.data[o.lo:o.hi], group=self.group)
