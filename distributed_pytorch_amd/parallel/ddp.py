"""Data-parallel engine: bucketed gradient all-reduce overlapped with the backward pass.

Gradients accumulate directly inside flat per-bucket buffers (see ``flat.py``). A
post-accumulate hook per parameter counts the bucket down; when the last gradient of a bucket
has landed, the whole bucket is all-reduced asynchronously on RCCL's stream while autograd
keeps computing earlier layers. Buckets are contiguous runs of transformer blocks, tens of MB
each (``bucket_mb``), because a full-mesh xGMI collective is bound by the per-link rate and
wants few large messages. The 1/world averaging is folded into the AdamW kernel
(``grad_mul``), the sum travels in the gradient dtype (bf16 on the GPU).

Differences from the reference's ``DistributedDataParallel`` use
(reference: multi-gpu/ddp/train.py:284-330, kaggle-ddp.py:1061-1105):

* no ``find_unused_parameters`` graph walk for MoE: buckets are zero-initialised flat buffers,
  so a bucket whose hooks never all fired (experts without tokens) is simply reduced when the
  backward pass has ended;
* gradient accumulation honours ``require_backward_grad_sync`` / ``no_sync()`` exactly like the
  reference loop toggles it;
* buffers are not re-broadcast from rank 0 on every forward; MoE routing biases are averaged
  after each step instead.

The tied ``lm_head`` / ``tkn_emb`` weight: its gradient has two contributions, the head's dense
``dlogits^T x`` (the FIRST thing backward produces) and the embedding scatter (the LAST), so its
bucket can only leave after backward has ended: the one collective of the step that nothing hides
(reference tie: single-gpu/model.py:559-560). Two ways to run it:

* ``tied_tail="fused"`` (default): one all-reduce of the summed gradient, in the gradient dtype —
  77 MB in bf16 for GPT-2-124M, half of what the reference's fp32 reducer sends;
* ``tied_tail="split"``: the head contribution is diverted into its own buffer and all-reduced as
  soon as the head's backward has run, under the whole rest of backward; the embedding part goes
  at the end and the two are added after both have arrived (all-reduce is linear). The exposed
  tail then carries the embedding scatter only. With a dense [V, C] buffer that is the same
  number of bytes as before while the job sends one extra buffer in total, so this only pays when
  the tail part travels cheaper than the head part (e.g. fp32 head gradients with
  ``grad_dtype="fp32"`` while the scatter part, which is exact in bf16 for up to 256 equal
  addends, is sent narrow) — hence opt-in.
"""
from __future__ import annotations

import torch

from . import comm
from .engine import Engine


class _DivertGrad(torch.autograd.Function):
    """Identity on the tied weight in forward; in backward the gradient that arrives (the head's dense
    contribution) goes to the engine's side buffer instead of the parameter."""

    @staticmethod
    def forward(ctx, w, engine):
        ctx.engine = engine
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        ctx.engine._head_grad_arrived(g)
        return None, None


class DDPEngine(Engine):
    name = "ddp"

    def _build_storage(self):
        super()._build_storage()
        self._head_sink = self._head_work = self._head_slot = None
        self.launch_log = []        # order of this step's gradient collectives ("head" or bucket index)
        if self.tied_tail == "split" and self._dist and type(self)._reduce_group is DDPEngine._reduce_group:
            self._setup_split_tail()
        self._pending = [0] * len(self.groups)
        self._launched = [False] * len(self.groups)
        self._works = []
        self._next = len(self.groups) - 1
        self._in_backward_sync = False
        for gi, g in enumerate(self.groups):
            for s in g.slots:
                s.param.register_post_accumulate_grad_hook(self._make_hook(gi))

    def _comm_buffers(self):
        # the gradient buckets (all-reduced / reduce-scattered in place) and, for the sharded-optimizer engines,
        # the parameter buffers their all-gather fills
        reduced = self.groups[:getattr(self, "_n_dense_groups", len(self.groups))]   # (expert groups are never reduced)
        bufs = [g.rgrad for g in reduced]
        if getattr(self, "stage", 0) in (1, 2):
            bufs += [g.data for g in reduced]
        if self._head_sink is not None:
            bufs.append(self._head_sink)
        return bufs

    def _setup_split_tail(self):
        head = getattr(self.module, "lm_head", None)
        emb = getattr(self.module, "tkn_emb", None)
        if head is None or emb is None or head.weight is not emb.weight:
            return
        for gi, g in enumerate(self.groups):
            for s in g.slots:
                if s.param is head.weight:
                    self._head_slot = (gi, s)
        if self._head_slot is None:
            return
        gi, s = self._head_slot
        self._head_sink = torch.zeros(s.numel, dtype=self.groups[gi].rgrad.dtype, device=self.device)
        self.module.head_weight_hook = lambda w: _DivertGrad.apply(w, self)

    def _head_grad_arrived(self, g: torch.Tensor):
        self._head_sink.add_(g.reshape(-1).to(self._head_sink.dtype))
        if self.require_backward_grad_sync:
            self.launch_log.append("head")
            self._head_work = comm.all_reduce(self._head_sink, group=self.group, async_op=True)

    def _make_hook(self, gi: int):
        def hook(param):
            self._pending[gi] -= 1
            if self._pending[gi] == 0 and self.require_backward_grad_sync and self._dist:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Buckets are reduced strictly from the last to the first (the order in which the backward
        pass completes them): a bucket goes out once it is complete AND every later bucket has gone.
        The sequence of collectives is then the same on every rank even if some rank's hooks fire in
        another order or not at all (such a bucket, and those before it, leave in finish_grad_sync)."""
        while self._next >= 0 and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next -= 1

    def _pre_forward(self):
        super()._pre_forward()
        for gi, g in enumerate(self.groups):
            self._pending[gi] = len(g.slots)
        if self.require_backward_grad_sync:
            self._launched = [False] * len(self.groups)
            self._next = len(self.groups) - 1
            self.launch_log = []

    # -- collectives ------------------------------------------------------------------------
    def _launch(self, gi: int):
        if self._launched[gi]:
            return
        self._launched[gi] = True
        self.launch_log.append(gi)
        self._works.append(self._reduce_group(gi))
        self._emulate_comm_kernels()

    def _emulate_comm_kernels(self):
        """DPA_EMULATE_COMM="<CUs>,<microseconds>" (benchmark aid, one GPU): with every bucket launch, hold
        that many CUs busy for that long on a side stream, as the kernels of a multi-GPU collective
        would, to measure how the backward kernels react to losing CUs."""
        import os
        spec = os.environ.get("DPA_EMULATE_COMM", "")
        if not spec or self.device.type != "cuda":
            return
        from .._ext import ops
        blocks, usec = (int(v) for v in spec.split(","))
        if not hasattr(self, "_emu"):
            self._emu = (torch.cuda.Stream(device=self.device), torch.zeros(1, device=self.device))
        side, sink = self._emu
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            ops().spin(blocks, usec, sink)

    def _reduce_group(self, gi: int):
        g = self.groups[gi]
        g.fold()
        return comm.all_reduce(g.rgrad, group=self.group, async_op=True)

    def finish_grad_sync(self):
        Engine.finish_grad_sync(self)
        if not self._dist:
            return
        if self.require_backward_grad_sync:
            # buckets whose hooks did not all fire (unused parameters) are reduced now, last first
            for gi in reversed(range(len(self.groups))):
                if not self._launched[gi]:
                    self._launch(gi)
        from ..utils.profiling import comm_wait_range
        with comm_wait_range():       # communication that backward did not hide
            for w in self._works:
                w.wait()
            if self._head_work is not None:
                self._head_work.wait()
        self._works = []
        if self._head_work is not None:
            gi, s = self._head_slot
            self.groups[gi].rgrad[s.offset:s.offset + s.numel].add_(self._head_sink)
            self._head_sink.zero_()
            self._head_work = None

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad()
        self._launched = [False] * len(self.groups)
        self._next = len(self.groups) - 1

    def _grad_mul(self) -> float:
        return 1.0 / self.world
