"""Training engines: the model + flat parameter storage + fused optimizer + gradient sync.

``Engine`` is the single-process form; ``parallel/ddp.py`` and ``parallel/zero.py`` subclass
it and only override *where gradients are reduced* and *which element range this rank
updates*; ``parallel/fsdp.py`` replaces the storage model (sharded parameters).

Lifecycle of one optimizer step (mirrors the reference loop, single-gpu/train.py:320-352):

    engine.zero_grad()
    for micro_step in range(accum):
        engine.require_backward_grad_sync = (micro_step == accum - 1)
        _, loss, _ = engine(x, y);  (loss / accum).backward()
    engine.clip_grad_norm_(max_norm)      # device-side norm; no host sync
    engine.step(lr)                       # fused AdamW (+ clip, + overflow skip) on flat state

Parameters are bf16 views of flat buffers on the GPU (fp32 on the CPU); the fp32 master
weights live in the optimizer. ``state_dict()`` returns fp32 tensors under the reference's
key schema whatever the sharding.
"""
from __future__ import annotations

import contextlib
from collections import OrderedDict

import torch
import torch.nn as nn

from . import comm
from .flat import FlatAdamW, FlatGroup, plan_groups


def resolve_dtype(name: str, device: torch.device) -> torch.dtype:
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("fp32", "float32"):
        return torch.float32
    return torch.bfloat16 if device.type == "cuda" else torch.float32


class _OptimizerFacade:
    """The slice of the torch optimizer interface the reference loop uses
    (``param_groups[i]['lr']``, ``zero_grad``, ``step``)."""

    def __init__(self, engine: "Engine", lr: float):
        self.engine = engine
        self.param_groups = [{"lr": lr, "weight_decay": engine.weight_decay},
                             {"lr": lr, "weight_decay": 0.0}]

    def zero_grad(self, set_to_none: bool = True):
        self.engine.zero_grad()

    def step(self):
        self.engine.step(self.param_groups[0]["lr"])

    def state_dict(self):
        return self.engine.optimizer_state_dict()

    def load_state_dict(self, sd):
        self.engine.load_optimizer_state_dict(sd)


class Engine(nn.Module):
    name = "single"

    def __init__(self, model: nn.Module, weight_decay: float = 0.1, lr: float = 3e-4, betas=(0.9, 0.999),
                 eps: float = 1e-8, dtype: str = "auto", bucket_mb: float = 48.0, device=None, group=None,
                 grad_dtype: str = "auto", vocab_parallel: bool = False, tied_tail: str = "fused",
                 ep_capacity: float = 0.0):
        super().__init__()
        self.ep_capacity = float(ep_capacity)   # expert-parallel engines: static per-peer capacity factor (0: exact)
        assert tied_tail in ("fused", "split"), tied_tail
        self.tied_tail = tied_tail
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.param_dtype = resolve_dtype(dtype, self.device)
        # "auto" / "bf16": gradients in the parameter dtype; "fp32": fp32 accumulation and reduction
        assert grad_dtype in ("auto", "bf16", "fp32"), grad_dtype
        self.grad_dtype = torch.float32 if grad_dtype == "fp32" else self.param_dtype
        self.module = model.to(self.device)
        self.group = group
        self.world = comm.world_size(group)
        self._dist = not comm._single(group)   # collectives are really issued (always when world > 1)
        if vocab_parallel and self._dist:
            V = getattr(getattr(model, "config", None), "vocab_size", 0)
            assert V and V >= self.world, f"--vocab_parallel: vocabulary {V} does not split over {self.world} ranks"
            model.vocab_parallel = (group,)
        self.rank = comm.get_rank(group)
        self.weight_decay = weight_decay
        self.betas, self.eps = betas, eps
        self.bucket_mb = bucket_mb
        self.require_backward_grad_sync = True
        self.step_count = 0
        self._max_norm = 0.0
        self._norm2 = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._have_norm = False
        # optimizer steps the AdamW kernel skipped because the gradient norm was not finite (device-side
        # count: the skip itself happens on the device, the host never looks at the norm)
        self._skipped = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._dyn = None   # device {lr, bc1, bc2_sqrt} while a captured step is recorded / replayed (parallel/graph.py)
        self._sync_initial_state()
        self._build_storage()
        self.registered_buffers = self._register_comm_buffers()
        self.optimizer = _OptimizerFacade(self, lr)

    # ---- construction ---------------------------------------------------------------------
    def _sync_initial_state(self):
        """Every rank starts from rank 0's weights and buffers (seeds differ per rank)."""
        if not self._dist:
            return
        with torch.no_grad():
            for t in list(self.module.parameters()) + [b for b in self.module.buffers() if b is not None]:
                if t.is_complex():
                    v = torch.view_as_real(t.data)
                    comm.broadcast(v, 0, self.group)
                else:
                    comm.broadcast(t.data, 0, self.group)

    def _bucket_elems(self) -> int:
        return int(self.bucket_mb * 1024 * 1024 / torch.empty((), dtype=self.param_dtype).element_size())

    def _pad_multiple(self) -> int:
        return 8 * self.world

    def _build_storage(self):
        plan = plan_groups(self.module, self._bucket_elems())
        self.groups = [FlatGroup(np_, self.param_dtype, self.device, self._pad_multiple(), self.weight_decay, name,
                                 grad_dtype=self.grad_dtype)
                       for name, np_ in plan]
        self.optims = [FlatAdamW(g, *self._opt_range(g), betas=self.betas, eps=self.eps) for g in self.groups]
        for g in self.groups:
            g.drop_init()
        self._enable_direct_grads()

    def _comm_buffers(self):
        """The persistent flat buffers this engine's collectives run on (same addresses for the whole run)."""
        return []

    def _register_comm_buffers(self) -> int:
        """With the framework's own RCCL communicator, register the persistent flat buffers once, so that RCCL can
        map them for direct peer access instead of staging every collective through its FIFO buffers. The library
        may decline a buffer (the collective then runs as before); registrations are released by
        ``comm.shutdown()`` / ``comm.deregister_buffers()``. Returns the number of accepted registrations."""
        if not self._dist or comm.own_comm() is None or self.device.type != "cuda":
            return 0
        return sum(1 for t in self._comm_buffers() if t is not None and t.numel() > 0 and comm.register_buffer(t))

    def _enable_direct_grads(self):
        """Let the weight-gradient kernel accumulate straight into the flat gradient buffers
        (see ``ops/linear.py``). Post-accumulate hooks still fire for such parameters."""
        import os
        if self.device.type != "cuda" or os.environ.get("DPA_DIRECT_GRAD", "1") != "1":
            return
        ln_params = set()
        if os.environ.get("DPA_DIRECT_LN", "1") == "1":
            ln_params = {id(p) for m in self.module.modules() if isinstance(m, nn.LayerNorm) for p in m.parameters()}
            ln_params |= {id(m.bias) for m in self.module.modules() if isinstance(m, nn.Linear) and m.bias is not None}
        for g in self.groups:
            for s in g.slots:
                # weight matrices; LayerNorm gamma / beta; Linear biases
                if s.param.dim() == 2 or id(s.param) in ln_params:
                    s.param._dpa_direct = True

    def _opt_range(self, g: FlatGroup):
        return 0, g.numel

    # The reference trainers set ``VAL_RUN`` on whatever wrapper they hold
    # (single-gpu/train.py:283); forward it to the real module.
    @property
    def VAL_RUN(self):
        return self.module.VAL_RUN

    @VAL_RUN.setter
    def VAL_RUN(self, v):
        self.module.VAL_RUN = v

    # ---- forward / backward ---------------------------------------------------------------
    def forward(self, *args, **kwargs):
        self._pre_forward()
        return self.module(*args, **kwargs)

    def _pre_forward(self):
        # fp32 gradient buffers: what the previous micro-step left in the bf16 views joins the fp32 sums
        # now, so accumulation over micro-steps never happens in bf16
        if self.grad_dtype != self.param_dtype and torch.is_grad_enabled():
            self._fold_grads()

    @contextlib.contextmanager
    def no_sync(self):
        prev = self.require_backward_grad_sync
        self.require_backward_grad_sync = False
        try:
            yield
        finally:
            self.require_backward_grad_sync = prev

    def zero_grad(self, set_to_none: bool = True):
        for g in self.groups:
            g.zero_grad()
        self._have_norm = False

    # ---- gradient sync hooks for subclasses -----------------------------------------------
    def finish_grad_sync(self):
        """Block the compute stream until every gradient collective has landed (and run whatever weight-gradient
        products an expert-parallel layer deferred and nobody flushed: none in a normal backward pass)."""
        from ..ops.linear import flush_deferred_wgrads
        flush_deferred_wgrads()

    def _grad_mul(self) -> float:
        return 1.0

    def _norm_ranges(self):
        return None

    def _reduce_norm(self):
        pass

    # ---- clipping + step ------------------------------------------------------------------
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Record the clip threshold and compute the global gradient norm on the device. The
        scaling itself happens inside the AdamW kernel. Returns the norm as a device tensor."""
        self.finish_grad_sync()
        self._max_norm = float(max_norm)
        self._compute_norm()
        return self._norm2.sqrt() * abs(self._grad_mul())

    def _fold_grads(self):
        """fp32 gradient buffers: fold what autograd left in the bf16 views (no-op otherwise)."""
        for g in self.groups:
            g.fold()

    def _compute_norm(self):
        from .flat import grad_norm_sq
        self._fold_grads()
        grad_norm_sq(self._norm_grads(), self._norm2, None)
        self._reduce_norm()
        self._have_norm = True

    def _norm_grads(self):
        return self.groups

    def step(self, lr: float | None = None):
        self.finish_grad_sync()
        if lr is None:
            lr = self.optimizer.param_groups[0]["lr"]
        if not self._have_norm:
            self._compute_norm()   # also provides the overflow check
        self.step_count += 1
        self._skipped += torch.isfinite(self._norm2).logical_not()
        self._apply_updates(float(lr))
        self._have_norm = False
        self._post_step()

    def skipped_steps(self) -> int:
        """Number of optimizer steps skipped so far for a non-finite gradient norm (reads the device:
        call it where the host synchronises anyway, e.g. when it reads the loss). The step counter that
        drives Adam's bias corrections keeps counting skipped steps: after t updates and s skips the
        corrections are those of step t + s, which only matters for the first few hundred steps."""
        return int(self._skipped.item())

    def _apply_updates(self, lr: float):
        for g, o in zip(self.groups, self.optims):
            o.step(g.rgrad[o.lo:o.hi], g.data[o.lo:o.hi], lr, self.step_count, self._norm2, self._max_norm,
                   self._grad_mul(), self._dyn)

    def _post_step(self):
        self._sync_moe_bias()

    def _sync_moe_bias(self):
        """Average the aux-free MoE routing biases over the data-parallel ranks, i.e. balance on
        the global expert load. (DDP in the reference re-broadcasts rank 0's buffer each forward,
        so only rank 0's load statistics survive there.)"""
        if not self._dist:
            return
        biases = [b for n, b in self.module.named_buffers() if n.endswith("expert_bias")]
        if not biases:
            return
        flat = torch.cat([b.reshape(-1).float() for b in biases])
        comm.all_reduce(flat, group=self.group)
        flat /= self.world
        off = 0
        for b in biases:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()

    # ---- reference-compatible helpers -----------------------------------------------------
    def configure_optimizers(self, weight_decay=None, learning_rate=None, device=None):
        if learning_rate is not None:
            for pg in self.optimizer.param_groups:
                pg["lr"] = learning_rate
        return self.optimizer

    def get_num_params(self):
        return self.module.get_num_params()

    # ---- checkpoint state -----------------------------------------------------------------
    def _full_master(self, gi: int) -> torch.Tensor:
        """fp32 master weights of a whole group (gathered if sharded)."""
        return self.optims[gi].master

    def state_dict(self, *args, **kwargs):  # noqa: D401 - torch signature
        """fp32 ``state_dict`` of the wrapped model in the reference's key schema."""
        sd = OrderedDict()
        by_param = {}
        for gi, g in enumerate(self.groups):
            full = self._full_master(gi)
            for s in g.slots:
                by_param[id(s.param)] = full[s.offset:s.offset + s.numel].view(s.shape).detach().clone()
        raw = self.module.state_dict()
        params = dict(self.module.named_parameters(remove_duplicate=False))
        for k, v in raw.items():
            p = params.get(k)
            sd[k] = by_param[id(p)] if p is not None and id(p) in by_param else v.detach().clone()
        return sd

    def load_model_state(self, sd: dict):
        """Load a reference-schema fp32 state dict into master + compute copies."""
        params = dict(self.module.named_parameters(remove_duplicate=False))
        with torch.no_grad():
            for gi, g in enumerate(self.groups):
                o = self.optims[gi]
                for s in g.slots:
                    key = next(k for k, p in params.items() if p is s.param and k in sd)
                    src = sd[key].to(self.device).reshape(-1)
                    g.data[s.offset:s.offset + s.numel].copy_(src)
                    lo, hi = max(s.offset, o.lo), min(s.offset + s.numel, o.hi)
                    if hi > lo:
                        o.master[lo - o.lo:hi - o.lo].copy_(src[lo - s.offset:hi - s.offset].float())
            for k, b in self.module.named_buffers():
                if k in sd and b is not None:
                    b.copy_(sd[k].to(b.device))

    def load_optimizer_state_dict(self, sd):
        assert sd["world"] == self.world, ("this resume file was written by a job of another world size: "
                                           "load_optimizer_shards() takes the files of all its ranks")
        self.step_count = sd["step"]
        for o, s in zip(self.optims, sd["groups"]):
            o.load_state_dict(s)
        with torch.no_grad():   # refresh the compute copies from the restored master weights
            self._refresh_params_from_master()

    supports_resharded_resume = True

    def _named_optimizer_group(self):
        """(FlatGroup, FlatAdamW) of the one group whose CONTENT depends on the world size (the local routed experts of
        the expert-parallel engines; always the last optimizer), or None. Such a group is saved with its slot names and
        restored by name; every other group is restored by position."""
        return None

    def optimizer_state_dict(self):
        sd = {"step": self.step_count, "rank": self.rank, "world": self.world,
              "groups": [o.state_dict() for o in self.optims]}
        named = self._named_optimizer_group()
        if named is not None:
            sd["named_slots"] = [(s.name, s.offset, s.numel) for s in named[0].slots]
        return sd

    def load_optimizer_shards(self, sds: list):
        """Restore from the optimizer states of ALL ranks of a job that ran with another world size (a restart on
        fewer or more GPUs). The flat groups are planned from the model alone, so a group's elements sit at the same
        offsets whatever the world size; only the slice [lo, hi) a rank owns and the zero padding behind the last
        parameter differ. Every element of this rank's slice is taken from the old shard that holds it. The group of
        local routed experts (expert-parallel engines) is restored tensor by tensor from whichever old rank owned it."""
        assert self.supports_resharded_resume, f"{self.name}: resume with a different world size is not supported"
        assert len(sds) == sds[0]["world"] and all(len(s["groups"]) == len(self.optims) for s in sds), \
            "the resume files do not belong to this model / engine layout"
        self.step_count = sds[0]["step"]
        named = self._named_optimizer_group()
        by_position = self.optims if named is None else self.optims[:-1]
        with torch.no_grad():
            for gi, o in enumerate(by_position):
                for name in ("master", "exp_avg", "exp_avg_sq"):
                    dst = getattr(o, name)
                    dst.zero_()                  # (elements no old shard covers are padding)
                    for s in sds:
                        og = s["groups"][gi]
                        lo, hi = max(og["lo"], o.lo), min(og["hi"], o.hi)
                        if hi > lo:
                            dst[lo - o.lo:hi - o.lo].copy_(og[name][lo - og["lo"]:hi - og["lo"]])
            if named is not None:
                group, optim = named
                where = {nm: (s["groups"][-1], off, n) for s in sds for nm, off, n in s.get("named_slots", [])}
                for slot in group.slots:
                    if slot.name not in where:
                        raise RuntimeError(f"the resume files hold no optimizer state for {slot.name}")
                    og, off, n = where[slot.name]
                    assert n == slot.numel, slot.name
                    for name in ("master", "exp_avg", "exp_avg_sq"):
                        getattr(optim, name)[slot.offset:slot.offset + slot.numel].copy_(og[name][off:off + n])
            self._refresh_params_from_master()

    def _refresh_params_from_master(self):
        for g, o in zip(self.groups, self.optims):
            g.data[o.lo:o.hi].copy_(o.master)
