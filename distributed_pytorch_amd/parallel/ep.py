"""Expert parallelism for the MoE layers: routed experts are sharded over the ranks of a
process group and tokens travel to their experts and back with two all-to-alls per layer.

Layout: with ``E`` ranks and ``n_routed`` routed experts, rank ``r`` owns the contiguous block
of ``n_routed / E`` experts starting at ``r * n_routed / E`` (8 experts on 8 GPUs -> one
expert per GPU, top-2 routing -> every token row is sent to at most two peers). On a fully
connected xGMI mesh every pair of GPUs has its own link, so an all-to-all with per-peer splits
uses all 7 links of every GPU at once.

Per MoE forward:
  1. route locally (same math as the single-device layer);
  2. sort the (token, slot) pairs by expert id == by destination rank;
  3. exchange the per-expert row counts (tiny int all-to-all), then the rows themselves (bf16);
  4. each rank regroups what it received by local expert, runs its experts on contiguous
     slices, and sends the outputs back along the reverse all-to-all;
  5. scale by the gate and scatter-add into the token order.
Every exchange is issued asynchronously and waited for right before its first consumer, in forward and
in backward (the backward of an exchange is the exchange with the split sizes swapped): the two halves of
the shared experts and the local experts' weight gradients run while the rows travel, see
``ExpertParallelDispatcher``.

Gradients: an expert receives every token routed to it from all ranks, so its gradient is
already the global sum — expert parameters are NOT all-reduced; dense parameters are reduced
as in ``DDPEngine``. Both are scaled by 1/W in the optimizer (the loss is a per-rank mean).
Non-local experts are dropped from the module's trainable set (their storage is freed);
``state_dict()`` gathers every expert back so that checkpoints keep the reference key schema.

The reference has the MoE layer but no expert parallelism (listed as a goal only,
reference: README.md:7; single-gpu/model.py:490-502 loops over all experts locally).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from .. import ops
from . import comm
from .ddp import DDPEngine
from .fsdp import FSDPEngine
from .flat import FlatAdamW, FlatGroup, unique_named_parameters


class _WaitGrad(torch.autograd.Function):
    """Identity. In backward the incoming gradient may be the output of an exchange that ``_Exchange.backward``
    issued and did not wait for: this node sits right in front of that gradient's first consumer and waits
    there. Everything autograd runs between the two nodes (the backward of whatever the forward computed while
    the forward exchange travelled: a half of the shared experts) runs under the exchange. ``open_defer``: the
    node in front of the local experts also opens the deferral window for their weight gradients."""

    @staticmethod
    def forward(ctx, x, box, key, open_defer):
        # ``box`` is the layer's mailbox (work handles in flight by key, the event log), NOT the dispatcher's state
        # dict: that one holds tensors whose grad_fn is this very node, and a node that held it back would close a
        # reference cycle through the autograd graph that nothing ever collects (gigabytes per step).
        ctx.box, ctx.key, ctx.open_defer = box, key, open_defer
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        pending = ctx.box.pop(ctx.key, None)
        if pending is not None:
            ctx.box["log"].append(("wait", ctx.key))
            pending[0].wait()
        if ctx.open_defer:
            from ..ops.linear import begin_deferred_wgrads
            begin_deferred_wgrads()
        return g, None, None, None


class _Exchange(torch.autograd.Function):
    """Autograd node of a row all-to-all(v) that was already issued on ``x`` (asynchronously, outside autograd)
    and waited for: forward hands out the buffer it filled. Backward issues the reverse exchange of the incoming
    gradient ASYNCHRONOUSLY and parks the work handle in ``st[key]`` for the ``_WaitGrad`` node in front of the
    gradient's consumer; with ``flush`` the weight gradients the local experts deferred run right behind the
    issue, i.e. while their input gradient is on the links."""

    @staticmethod
    def forward(ctx, x, out, in_splits, out_splits, group, box, key, flush):
        ctx.meta = (tuple(x.shape), in_splits, out_splits, group, box, key, flush)   # box: see _WaitGrad
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        xshape, in_splits, out_splits, group, st, key, flush = ctx.meta
        gx = g.new_empty(xshape)
        src = g.contiguous()
        w = comm.all_to_all(gx, src, in_splits, out_splits, group=group, async_op=True)
        st[key] = (w, src)                   # the source stays alive until the wait
        st["log"].append(("issue", key))
        if flush:
            from ..ops.linear import flush_deferred_wgrads
            n = flush_deferred_wgrads()
            st["log"].append(("wgrads", n))
        return gx, None, None, None, None, None, None, None


class ExpertParallelDispatcher:
    """Installed as ``MoE.dispatcher``. Three phases, so that both exchanges of a layer travel under work that does
    not depend on them:

      ``begin``  routes, sorts and issues the dispatch exchange asynchronously        | first half of the shared experts
      ``mid``    waits for it, runs the local experts, issues the return exchange     | second half of the shared experts
      ``finish`` waits for it and combines.

    Backward mirrors it through the autograd nodes above: the reverse of the return exchange is issued, the
    second shared half's backward runs, the wait comes right before the local experts' backward; their weight
    gradients are deferred, so the reverse of the dispatch exchange leaves as soon as their INPUT gradient exists
    and travels under the weight-gradient GEMMs and the first shared half's backward.

    ``capacity`` = 0 (default): exact. The per-peer row counts are read back to the host once per layer
    (they are the split sizes of a variable all-to-all) and every routed pair is served.
    ``capacity`` = F > 0: static shapes and NO host read. Every rank sends exactly
    ``ceil(F * pairs / W)`` rows to every peer (zero rows past the real ones); pairs beyond a peer's
    capacity are dropped, i.e. contribute nothing for that expert, and are counted on the device in
    ``dropped`` (``F >= W`` can never drop). The whole layer is then enqueued without the host ever
    waiting for the GPU, which is what keeps the kernel queue full across MoE layers."""

    def __init__(self, group=None, capacity: float = 0.0):
        self.group = group
        self.world = comm.world_size(group)
        self.rank = comm.get_rank(group)
        self.capacity = float(capacity)
        self.dropped = None        # device counter of pairs dropped by the capacity bound
        self.last_log = None       # event log of the most recent layer (forward and backward), for tests / debugging

    def local_expert_range(self, n_routed: int):
        assert n_routed % self.world == 0, "n_routed must be divisible by the expert-parallel world size"
        per = n_routed // self.world
        return self.rank * per, (self.rank + 1) * per

    def dropped_pairs(self) -> int:
        return int(self.dropped.item()) if self.dropped is not None else 0

    def __call__(self, moe, x_flat, topk_indices, topk_gates):
        return self.finish(self.mid(self.begin(moe, x_flat, topk_indices, topk_gates)))

    def _issue(self, st, src, n_out, in_splits, out_splits, key):
        """Asynchronous exchange of ``src`` (outside autograd) -> (buffer, work handle)."""
        buf = src.new_empty((n_out,) + tuple(src.shape[1:]))
        w = comm.all_to_all(buf, src.detach().contiguous(), out_splits, in_splits, group=self.group, async_op=True)
        st["log"].append(("issue", key))
        return buf, w

    # ---- phase 1 --------------------------------------------------------------------------------
    def begin(self, moe, x_flat, topk_indices, topk_gates):
        W, per = self.world, moe.n_routed // self.world
        N, k = topk_indices.shape
        # expert-sorted (== destination-rank-sorted) send order, computed on the device
        row_of_pair, counts = ops.plan_dense(topk_indices, moe.n_routed)   # counts: rows per expert (global ids)
        box = {"log": []}            # what the autograd nodes of this layer share: pending work handles and the log
        st = dict(moe=moe, x_flat=x_flat, row_of_pair=row_of_pair, gates=topk_gates, k=k, per=per, log=box["log"], box=box)
        self.last_log = st["log"]
        if self.capacity > 0:
            cap = -(-int(self.capacity * N * k) // W)
            cap = max(8, (cap + 7) // 8 * 8)
            cnt = counts.view(W, per)
            want = cnt.sum(1)                                            # rows each destination should get
            seg_start = torch.cumsum(want, 0) - want
            end = torch.cumsum(cnt, 1)
            send_cnt = end.clamp(max=cap) - (end - cnt).clamp(max=cap)   # per (destination, expert), clipped
            sent = send_cnt.sum(1)
            lost = (want - sent).sum()
            self.dropped = lost if self.dropped is None else self.dropped + lost
            # The rows go from x straight into the padded send layout (``cap`` rows per destination): the pair at
            # place j of its destination's expert-sorted run lands in row dest * cap + j, a pair past the capacity
            # gets no row (-1), and the rows no pair fills are the only ones zeroed. Index arithmetic on [pairs]
            # integers; the activations are moved once, by the scatter kernel.
            dest = torch.div(topk_indices.reshape(-1), per, rounding_mode="floor")
            j = row_of_pair - seg_start[dest]
            row_cap = torch.where(j < cap, dest * cap + j, torch.full_like(j, -1))
            lanes = torch.arange(W, device=x_flat.device, dtype=torch.int64)
            tables = (lanes * cap, sent.to(torch.int64).contiguous(), torch.full_like(lanes, cap))
            send = ops.scatter_rows(x_flat, row_cap, k, W * cap, tables)
            recv_cnt = torch.empty_like(send_cnt)
            comm.all_to_all(recv_cnt, send_cnt.contiguous(), group=self.group)   # [src rank][local expert]
            st.update(cap=cap, row_cap=row_cap, tables=tables, recv_cnt=recv_cnt, in_splits=None, out_splits=None)
            n_out = W * cap
        else:
            send = ops.scatter_rows(x_flat, row_of_pair, k, N * k)              # enqueued ahead of the host read below
            recv_counts = torch.empty_like(counts)
            comm.all_to_all(recv_counts, counts, group=self.group)              # [src rank][local expert]
            recv_mat = recv_counts.view(W, per)
            # the one host read of the layer: rows sent to / received from each rank (the all-to-all splits)
            counts_h, recv_h = torch.stack([counts.view(W, per).sum(1), recv_mat.sum(1)]).tolist()
            counts_h, recv_h = [int(v) for v in counts_h], [int(v) for v in recv_h]
            st.update(recv_mat=recv_mat, in_splits=counts_h, out_splits=recv_h)
            n_out = sum(recv_h)
        send_w = _WaitGrad.apply(send, box, "dispatch_bwd", False)
        st["send_w"] = send_w
        st["recv_buf"], st["work"] = self._issue(st, send_w, n_out, st["in_splits"], st["out_splits"], "dispatch")
        return st

    # ---- phase 2 --------------------------------------------------------------------------------
    def _local_experts(self, moe, recv, row_expert, row_weight, per):
        """Outputs of this rank's experts for the received rows (row_weight 0 marks padding rows)."""
        lo = self.rank * per
        local = [moe.experts[moe.n_shared + lo + j].expert for j in range(per)]
        if per == 1:
            # one expert per rank (e.g. 8 experts on the 8 GPUs of a node): everything received is for
            # that expert, so it is one dense MLP over the rows as they arrived — no regrouping at all.
            # Padding rows of the static-capacity exchange are masked by the sender on return.
            return local[0](recv)
        fcs, projs = [m.c_fc.weight for m in local], [m.c_proj.weight for m in local]
        act = local[0].non_linearity
        if ops.grouped_supported(recv, fcs, projs, act):
            # grouped MFMA GEMMs over the padded expert-sorted layout: one launch per product for all
            # local experts, nothing read back to the host (dropout on the experts' outputs included)
            p_drop = moe.config.dropout if moe.training else 0.0
            return ops.grouped_experts(recv, row_expert.view(-1, 1), row_weight.view(-1, 1).to(recv.dtype), fcs,
                                       projs, act, p_drop=p_drop)
        return None

    def mid(self, st):
        moe, per, W = st["moe"], st["per"], self.world
        st.pop("work").wait()
        st["log"].append(("wait", "dispatch"))
        recv = _Exchange.apply(st["send_w"], st.pop("recv_buf"), st["in_splits"], st["out_splits"], self.group, st["box"],
                               "dispatch_bwd", True)
        R = recv.shape[0]
        if self.capacity > 0:
            cap = st["cap"]
            ends = torch.cumsum(st["recv_cnt"], 1)                            # [src, per]
            j = torch.arange(cap, device=recv.device).expand(W, cap).contiguous()
            row_expert = torch.searchsorted(ends, j, right=True).clamp(max=per - 1).reshape(-1)
            row_valid = (j < ends[:, -1:]).reshape(-1)
            y_recv = self._local_experts(moe, recv, row_expert, row_valid, per)
            if y_recv is None:                                                # eager: every expert sees every row, masked
                lo = self.rank * per
                y_recv = torch.zeros_like(recv)
                for e in range(per):
                    m = ((row_expert == e) & row_valid)[:, None].to(recv.dtype)
                    y_recv = y_recv + moe.experts[moe.n_shared + lo + e].expert(recv) * m
            n_back = W * cap
        else:
            recv_mat = st["recv_mat"]
            # regroup the received rows (ordered by source rank, then expert) by local expert
            seg_expert = torch.arange(per, device=recv.device).repeat(W)        # expert id of each (src, expert) segment
            row_expert = torch.repeat_interleave(seg_expert, recv_mat.reshape(-1), output_size=R)
            y_recv = self._local_experts(moe, recv, row_expert, torch.ones(R, device=recv.device), per)
            if y_recv is None:
                lo = self.rank * per
                local = [moe.experts[moe.n_shared + lo + j].expert for j in range(per)]
                perm = torch.argsort(row_expert, stable=True)
                per_expert = recv_mat.sum(0).tolist()
                grouped = recv[perm]
                outs, start = [], 0
                for j, c in enumerate(per_expert):
                    c = int(c)
                    if c:
                        outs.append(local[j](grouped[start:start + c]))
                    start += c
                if outs:
                    y_grouped = torch.cat(outs, dim=0)
                else:
                    y_grouped = grouped.new_zeros((0, st["x_flat"].shape[1]))
                    # keep the local experts in the autograd graph so their (zero) gradients exist
                    y_grouped = y_grouped + 0.0 * sum(p.sum() for m in local for p in m.parameters())
                y_recv = torch.empty_like(y_grouped).index_copy_(0, perm, y_grouped) if R else y_grouped
            n_back = sum(st["in_splits"])
        # the node in front of the local experts' backward: waits for the reverse of the return exchange and opens
        # the deferral window for the experts' weight gradients
        y_w = _WaitGrad.apply(y_recv, st["box"], "return_bwd", True)
        st["y_w"] = y_w
        st["back_buf"], st["work"] = self._issue(st, y_w, n_back, st["out_splits"], st["in_splits"], "return")
        return st

    # ---- phase 3 --------------------------------------------------------------------------------
    def finish(self, st):
        if "y_w" not in st:
            st = self.mid(st)
        st.pop("work").wait()
        st["log"].append(("wait", "return"))
        y_back = _Exchange.apply(st.pop("y_w"), st.pop("back_buf"), st["out_splits"], st["in_splits"], self.group, st["box"],
                                 "return_bwd", False)
        if self.capacity > 0:
            # padding rows of the returned buffer are never read: every pair gathers its own row, a dropped pair none
            return ops.gather_combine(y_back, st["row_cap"], st["gates"], st["k"], st["tables"]).to(st["x_flat"].dtype)
        return ops.gather_combine(y_back, st["row_of_pair"], st["gates"], st["k"]).to(st["x_flat"].dtype)


def _expert_index(name: str):
    """('transformer.h.3.moe.experts.5.expert.c_fc.weight') -> 5, or None."""
    parts = name.split(".")
    if "experts" in parts:
        i = parts.index("experts")
        if i > 0 and parts[i - 1] == "moe":
            return int(parts[i + 1])
    return None


class EPEngine(DDPEngine):
    """Data parallel for the dense parameters + expert parallel for the routed experts, both
    over the same process group."""

    name = "ep"

    def _named_optimizer_group(self):
        # which experts a rank owns (and keeps optimizer state for) depends on the world size
        return (self.groups[-1], self.optims[-1]) if len(self.groups) > self._n_dense_groups else None

    def _build_storage(self):
        self.dispatcher = ExpertParallelDispatcher(self.group, capacity=self.ep_capacity)
        cfg = self.module.config
        assert cfg.moe, "EPEngine needs an MoE model"
        local_named = _drop_remote_experts(self, cfg)
        # dense parameters: bucketed + all-reduced (DDP); local experts: one group, never reduced
        for _, p in local_named:
            p.requires_grad_(False)
        super()._build_storage()
        for _, p in local_named:
            p.requires_grad_(True)
        self._n_dense_groups = len(self.groups)
        if local_named:
            g = FlatGroup(local_named, self.param_dtype, self.device, 8, self.weight_decay, "experts",
                          grad_dtype=self.grad_dtype)
            self.groups.append(g)
            self.optims.append(FlatAdamW(g, 0, g.numel, betas=self.betas, eps=self.eps))
            g.drop_init()
            self._pending.append(0)
            self._launched.append(True)
            self._enable_direct_grads()   # the base class marked its groups before this one existed

    def _skip_expert_groups(self):
        """Expert gradients are complete locally: nothing to reduce, and no hook counts them down. The
        last-to-first launch order of the dense buckets therefore starts below them (left at the expert
        group, whose count never reaches zero, it would hold every dense all-reduce back until
        finish_grad_sync and lose the overlap with backward)."""
        for gi in range(self._n_dense_groups, len(self.groups)):
            self._launched[gi] = True
            self._pending[gi] = 0
        self._next = self._n_dense_groups - 1

    def _pre_forward(self):
        super()._pre_forward()
        self._skip_expert_groups()

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad()
        self._skip_expert_groups()

    def _compute_norm(self):
        """||g||^2 = dense part (replicated, counted once) + sum over ranks of the expert parts."""
        dense = torch.zeros_like(self._norm2)
        first = True
        self._fold_grads()
        for g in self.groups[:self._n_dense_groups]:
            ops.sumsq(g.rgrad, dense, accumulate=not first)
            first = False
        expert = torch.zeros_like(self._norm2)
        first = True
        for g in self.groups[self._n_dense_groups:]:
            ops.sumsq(g.rgrad, expert, accumulate=not first)
            first = False
        comm.all_reduce(expert, group=self.group)
        self._norm2.copy_(dense + expert)
        self._have_norm = True

    def state_dict(self, *args, **kwargs):
        """Reference-schema state dict with every expert gathered from its owner (collective)."""
        sd = OrderedDict()
        by_name = {}
        for gi, g in enumerate(self.groups):
            full = self._full_master(gi)
            for s in g.slots:
                by_name[s.name] = full[s.offset:s.offset + s.numel].view(s.shape).detach().clone()
        cfg = self.module.config
        per = (cfg.n_exp - cfg.n_shared) // self.world
        names = [k for k, _ in self.module.named_parameters(remove_duplicate=False)]
        params = dict(self.module.named_parameters(remove_duplicate=False))
        first_name = {}
        for k, p in params.items():
            first_name.setdefault(id(p), k)
        for k in names:
            ei = _expert_index(k)
            if ei is not None and ei >= cfg.n_shared:
                owner = (ei - cfg.n_shared) // per
                shape = self._expert_shape(k)
                t = by_name[k].to(self.device) if owner == self.rank else torch.empty(shape, device=self.device)
                comm.broadcast(t, src=owner, group=self.group)
                sd[k] = t
            else:
                sd[k] = by_name[first_name[id(params[k])]]
        for k, b in self.module.state_dict().items():
            if k not in sd and k not in params:
                sd[k] = b.detach().clone()
        order = [k for k in self.module.state_dict().keys() if k in sd]
        return OrderedDict((k, sd[k]) for k in order)

    def _expert_shape(self, name: str):
        cfg = self.module.config
        if name.endswith("c_fc.weight"):
            return ((2 if cfg.non_linearity.lower() == "swiglu" else 1) * cfg.up_dim, cfg.n_embd)
        return (cfg.n_embd, cfg.up_dim)


def _drop_remote_experts(engine, cfg):
    """Shared by the expert-parallel engines: install the dispatcher, free the routed experts that
    other ranks own and return the (name, parameter) list of the local ones."""
    n_routed = cfg.n_exp - cfg.n_shared
    lo, hi = engine.dispatcher.local_expert_range(n_routed)
    engine._local_routed = (lo, hi)
    local_named, engine._dropped = [], []
    for blk in engine.module.transformer.h:
        blk.moe.dispatcher = engine.dispatcher
    for n, p in unique_named_parameters(engine.module):
        ei = _expert_index(n)
        if ei is not None and ei >= cfg.n_shared:
            if lo <= ei - cfg.n_shared < hi:
                local_named.append((n, p))
                # never reduced across ranks and no bucket accounting waits for it: its weight-gradient product
                # may run after the exchange of the experts' input gradient has been issued (ops/linear.py)
                p._dpa_defer = True
            else:
                p.requires_grad_(False)
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)   # not owned here: free it
                engine._dropped.append(n)
    return local_named


class EPFSDPEngine(FSDPEngine):
    """Expert parallel for the routed experts + fully-sharded data parallel for everything else,
    over the same process group: dense parameters (attention, norms, shared experts, router,
    embeddings) live in per-block FSDP units that are all-gathered / reduce-scattered as in
    ``FSDPEngine``; each rank's own routed experts stay resident, receive their tokens through
    the all-to-all dispatcher and are never reduced (their gradient is already the global sum)."""

    name = "ep_fsdp"

    def _named_optimizer_group(self):
        return (self.expert_group, self.expert_optim) if self.expert_group is not None else None

    def _build_storage(self):
        self.dispatcher = ExpertParallelDispatcher(self.group, capacity=self.ep_capacity)
        cfg = self.module.config
        assert cfg.moe, "EPFSDPEngine needs an MoE model"
        local_named = _drop_remote_experts(self, cfg)
        for _, p in local_named:       # keep the experts out of the sharded units
            p.requires_grad_(False)
        super()._build_storage()
        for _, p in local_named:
            p.requires_grad_(True)
        self.expert_group = self.expert_optim = None
        if local_named:
            g = FlatGroup(local_named, self.param_dtype, self.device, 8, self.weight_decay, "experts",
                          grad_dtype=self.grad_dtype)
            self.expert_group = g
            self.expert_optim = FlatAdamW(g, 0, g.numel, betas=self.betas, eps=self.eps)
            self.optims.append(self.expert_optim)    # checkpointed with the unit optimizers
            g.drop_init()
            if self.device.type == "cuda":
                for s in g.slots:
                    if s.param.dim() == 2:
                        s.param._dpa_direct = True

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad()
        if self.expert_group is not None:
            self.expert_group.zero_grad()

    def _compute_norm(self):
        first = True
        for u in self.units:
            ops.sumsq(u.grad_shard, self._norm2, accumulate=not first)
            first = False
        if self.expert_group is not None:
            self.expert_group.fold()
            ops.sumsq(self.expert_group.rgrad, self._norm2, accumulate=True)
        comm.all_reduce(self._norm2, group=self.group)
        self._have_norm = True

    def _apply_updates(self, lr: float):
        super()._apply_updates(lr)
        if self.expert_group is not None:
            g, o = self.expert_group, self.expert_optim
            o.step(g.rgrad, g.data, lr, self.step_count, self._norm2, self._max_norm, self._grad_mul(), self._dyn)

    def _refresh_params_from_master(self):
        if self.expert_group is not None:
            self.expert_group.data.copy_(self.expert_optim.master)
        super()._refresh_params_from_master()

    def state_dict(self, *args, **kwargs):
        """Reference-schema fp32 state dict: dense tensors gathered from the shards, every routed
        expert broadcast from its owner (collective: call on every rank)."""
        by_param = {}
        for gi, u in enumerate(self.units):
            full = self._full_master(gi)
            for s in u.slots:
                by_param[id(s.param)] = full[s.offset:s.offset + s.numel].view(s.shape).clone()
        if self.expert_group is not None:
            for s in self.expert_group.slots:
                by_param[id(s.param)] = self.expert_optim.master[s.offset:s.offset + s.numel].view(s.shape).clone()
        cfg = self.module.config
        per = (cfg.n_exp - cfg.n_shared) // self.world
        sd = OrderedDict()
        for k, p in self.module.named_parameters(remove_duplicate=False):
            ei = _expert_index(k)
            if ei is not None and ei >= cfg.n_shared:
                owner = (ei - cfg.n_shared) // per
                t = by_param[id(p)].to(self.device) if owner == self.rank else \
                    torch.empty(EPEngine._expert_shape(self, k), device=self.device)
                comm.broadcast(t, src=owner, group=self.group)
                sd[k] = t
            else:
                sd[k] = by_param[id(p)]
        for k, b in self.module.named_buffers():
            mod_path, _, leaf = k.rpartition(".")
            owner_mod = self.module.get_submodule(mod_path) if mod_path else self.module
            if b is not None and leaf not in owner_mod._non_persistent_buffers_set:
                sd[k] = b.detach().clone()
        order = [k for k in self._state_dict_keys() if k in sd]
        return OrderedDict((k, sd[k]) for k in order)

    def load_model_state(self, sd: dict):
        if self.expert_group is not None:
            params = dict(self.module.named_parameters(remove_duplicate=False))
            with torch.no_grad():
                for s in self.expert_group.slots:
                    key = next(k for k, p in params.items() if p is s.param and k in sd)
                    src = sd[key].to(self.device).reshape(-1)
                    self.expert_group.data[s.offset:s.offset + s.numel].copy_(src)
                    self.expert_optim.master[s.offset:s.offset + s.numel].copy_(src.float())
        super().load_model_state(sd)
