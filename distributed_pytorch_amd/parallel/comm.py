"""Process-group bootstrap and the collective entry points of the framework.

One process per GPU, launched by ``torchrun`` (env ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` /
``MASTER_ADDR`` / ``MASTER_PORT``). Every collective of the framework goes through this module,
operates on large flat buffers and is asynchronous by default, so it is the single place to count
bytes, switch backends, or tune RCCL. Three transports sit behind the same functions:

* ``nccl``  — c10d's ProcessGroupNCCL, which on ROCm *is* RCCL over xGMI (the default on the GPU);
* ``rccl``  — the framework's own communicator (csrc/comm/rccl_comm.cpp): an ``ncclComm_t`` created
  from a unique id that rank 0 publishes over a small gloo group, its own HIP stream with an explicit
  priority (high / default / low), per-communicator channel bounds, buffer registration and
  per-collective device timings. GPU tensors take this path; host tensors and Python objects
  (checkpoint bookkeeping) use the gloo group that bootstrapped it;
* ``gloo``  — CPU plumbing tests, and several ranks rehearsing on one GPU.

The reference never calls a collective itself: DDP's reducer, ``ZeroRedundancyOptimizer`` and
FSDP issue them implicitly (reference: multi-gpu/ddp/train.py:19-36; kaggle-zero1.py:1062-1077;
kaggle-fsdp.py:1061-1086).

RCCL notes for MI355X (8 GPUs fully connected by 7 xGMI links each): a collective over the
full mesh is bound by the per-link rate, so few large messages (tens of MB) are preferred over
many small ones, and bf16 transport halves link time. The defaults in ``configure_rccl_env``
leave RCCL's topology detection alone and only set what is needed for this pool.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


# DPA_FORCE_DIST=1: create the process group and issue every collective even with one rank. A
# single-GPU box can then rehearse the RCCL call paths (in-place reduce-scatter / all-gather views,
# async work handles, stream ordering) exactly as a multi-rank job issues them.
FORCE_COLLECTIVES = os.environ.get("DPA_FORCE_DIST", "0") == "1"


def _single(group=None) -> bool:
    return world_size(group) == 1 and not (FORCE_COLLECTIVES and dist.is_initialized())


@dataclass
class DistInfo:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1
    device: torch.device = torch.device("cpu")
    backend: str = "none"

    @property
    def is_master(self) -> bool:
        return self.rank == 0


_SETTINGS = {"stream_priority": "default", "backend": None, "min_ctas": 0, "max_ctas": 0}
_OWN = None          # torch.classes.dpa.RcclComm of the default group when the backend is "rccl"
_PRIORITIES = {"high": -1, "default": 0, "low": 1}


def configure_rccl_env(max_channels: int | None = None, min_channels: int | None = None,
                       stream_priority: str | None = None, backend: str | None = None):
    """Environment of the RCCL communicator; call before ``init_distributed``.

    ``max_channels`` / ``min_channels`` (> 0) bound the number of RCCL channels, i.e. of workgroups
    (CUs) a collective's kernel occupies beside the backward pass: on the full xGMI mesh a handful of
    channels per peer saturates the links, and every channel beyond that only takes CUs away from the
    MFMA kernels. ``stream_priority`` ("default" | "high") is the priority of the HIP stream the
    collectives run on (high, c10d transport only: a ready collective's kernel is scheduled ahead of queued
    compute kernels, so a bucket leaves as soon as it is complete; see ``check_own_priority`` for why the own
    communicator takes the default priority only). ``backend`` ("nccl" | "rccl" | "gloo") selects the
    transport, see the module docstring; with "rccl" the channel bounds are also given to that one
    communicator as its CTA bounds instead of to every communicator of the process. Arguments left at None
    keep what an earlier call set (``init_distributed`` calls this once more with no arguments)."""
    # dmabuf IPC is the only mode the host driver supports on this pool.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Keep the rendezvous on the loopback interface: container hostnames may not resolve.
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if max_channels is not None:
        if max_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = str(int(max_channels))
        _SETTINGS["max_ctas"] = max(int(max_channels), 0)
    if min_channels is not None:
        if min_channels > 0:
            os.environ["NCCL_MIN_NCHANNELS"] = str(int(min_channels))
        _SETTINGS["min_ctas"] = max(int(min_channels), 0)
    if stream_priority is not None:
        assert stream_priority in _PRIORITIES, stream_priority
        _SETTINGS["stream_priority"] = stream_priority
    if backend is not None:
        assert backend in ("nccl", "rccl", "gloo", "c10d", "own"), backend
        _SETTINGS["backend"] = {"c10d": "nccl", "own": "rccl"}.get(backend, backend)


def rccl_settings() -> dict:
    """What the communicator was (or will be) configured with: for result records."""
    return {"NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
            "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
            "stream_priority": _SETTINGS["stream_priority"],
            "backend": "rccl(own)" if _OWN is not None else (_backend() if dist.is_initialized() else "none"),
            "rccl_version": _OWN.rccl_version() if _OWN is not None else None,
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}


def arm_hang_dump():
    """DPA_HANG_DUMP_S=<seconds>: every that many seconds each rank writes the Python stack of all its
    threads to stderr. A job stuck in a collective then shows, per rank, which call it is waiting in."""
    secs = float(os.environ.get("DPA_HANG_DUMP_S", "0") or 0)
    if secs > 0:
        import faulthandler
        faulthandler.dump_traceback_later(secs, repeat=True)


def init_distributed(backend: str | None = None, timeout_s: int = 600) -> DistInfo:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    arm_hang_dump()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # DPA_BACKEND=gloo lets several ranks share one GPU (rehearsal of the GPU code paths of the
    # engines on a single-GPU box; RCCL itself refuses two ranks on one device).
    backend = backend or os.environ.get("DPA_BACKEND") or _SETTINGS["backend"] or None
    backend = {"c10d": "nccl", "own": "rccl"}.get(backend, backend)
    ndev = torch.cuda.device_count() if use_cuda else 0
    device = torch.device(f"cuda:{local_rank % ndev}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world == 1 and not FORCE_COLLECTIVES:
        return DistInfo(0, 0, 1, device, "none")
    configure_rccl_env()
    backend = backend or ("nccl" if use_cuda else "gloo")
    if backend == "rccl" and not use_cuda:
        backend = "gloo"
    if not dist.is_initialized():
        # the own communicator is bootstrapped (and host-side objects are exchanged) over a gloo group
        kwargs = dict(backend="gloo" if backend == "rccl" else backend, rank=rank, world_size=world,
                      timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl":
            kwargs["device_id"] = device
            if _SETTINGS["stream_priority"] == "high":
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kwargs["pg_options"] = opts
        restart = int(os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") or 0)
        if restart > 0:
            # Relaunched by torchrun --max-restarts: the agent's key-value store still holds the dead
            # attempt's transport addresses under the default keys; give this attempt its own namespace.
            base, _, _ = next(dist.rendezvous("env://", rank, world, timeout=kwargs["timeout"]))
            kwargs["store"] = dist.PrefixStore(f"dpa_attempt_{restart}", base)
        dist.init_process_group(**kwargs)
    if backend == "rccl":
        _create_own(rank, world, device)
    return DistInfo(rank, local_rank, world, device, backend)


def _create_own(rank: int, world: int, device: torch.device):
    """Rank 0 draws the unique id, everyone receives it over gloo, every rank joins the communicator."""
    global _OWN
    if _OWN is not None:
        return
    from .. import _ext
    assert _ext.available(), "the own RCCL communicator lives in the HIP extension (python build.py)"
    cls = torch.classes.dpa.RcclComm
    check_own_priority(_SETTINGS["stream_priority"])
    uid = cls.unique_id() if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    dist.broadcast(uid, src=0)
    _OWN = cls(uid, rank, world, device.index or 0, _PRIORITIES[_SETTINGS["stream_priority"]],
               _SETTINGS["min_ctas"], _SETTINGS["max_ctas"])
    seen = _OWN.comm_count()
    assert seen == world, f"RCCL reports {seen} ranks in the communicator, the launcher {world}"


def check_own_priority(priority: str) -> None:
    """The own communicator runs on a default-priority stream only. Measured on ROCm 7.0.2 / RCCL 2.26.6 with one
    rank (benchmarks/debug/prio_diag.sh, profiles/r3_own_comm_priority_diagnosis.txt): RCCL calls issued on a
    high-priority pool stream or on a hand-made low-priority stream stop completing after one to five calls, at a
    call that changes from run to run; re-recording one event or a fresh one per call, explicit group calls or
    none, and stream queries after the enqueue make no difference; the same event chain without RCCL calls and
    c10d's high-priority option (``--comm_backend c10d --comm_priority high``) run. There is no override: a
    priority that may hang a job at a random step is not an option to offer."""
    if priority != "default":
        raise ValueError(f"--comm_priority {priority} is not usable under the own RCCL communicator on this ROCm "
                         "stack (collectives stop completing after a few calls); use the default priority, or "
                         "--comm_backend c10d for a high-priority stream")


def own_comm():
    """The framework-owned RCCL communicator of the default group (None on the c10d / gloo transports)."""
    return _OWN


def register_buffer(t: torch.Tensor) -> bool:
    """Register a persistent flat buffer with the own communicator (no-op elsewhere)."""
    return bool(_OWN.register_buffer(t)) if _OWN is not None and t.is_cuda else False


def registered_buffers() -> int:
    """Number of buffers currently registered with the own communicator."""
    return int(_OWN.num_registered()) if _OWN is not None else 0


def deregister_buffers() -> int:
    """Release every registration (after the communication stream has drained); returns how many there were.
    ``shutdown()`` does the same before it destroys the communicator."""
    return int(_OWN.deregister_buffers()) if _OWN is not None else 0


def set_comm_timing(on: bool):
    """Device-side start/stop events around every collective of the own communicator."""
    if _OWN is not None:
        _OWN.set_timing(bool(on))


def drain_comm_timings():
    """[(kind, bytes, ms on the communication stream)] of the finished collectives since the last call."""
    return list(_OWN.drain_timings()) if _OWN is not None else []


def check_comm_health():
    """Raise if RCCL has recorded an asynchronous error on the own communicator (a peer died, a link
    failed): called at the logging points of the training loop so that a broken job stops instead of hanging."""
    if _OWN is not None:
        err = _OWN.async_error()
        if err:
            raise RuntimeError(f"RCCL communicator error: {err}")


def shutdown():
    global _OWN
    if _OWN is not None:
        _OWN.destroy()
        _OWN = None
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


_OPS = {}


def _own_op(op) -> int:
    if not _OPS:
        R = dist.ReduceOp
        _OPS.update({R.SUM: 0, R.PRODUCT: 1, R.MAX: 2, R.MIN: 3, R.AVG: 4})
    return 0 if op is None else _OPS[op]


def _use_own(t: torch.Tensor, group) -> bool:
    if _OWN is None or not t.is_cuda:
        return False
    assert group is None, "the own RCCL communicator spans the default group only"
    return True


def _own_done(work, async_op: bool):
    """Synchronous collectives order the caller's stream behind the collective on the device (the host
    does not block), as c10d's do."""
    if async_op:
        return work
    work.wait()
    return _Done()


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def get_rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def _backend(group=None) -> str:
    return dist.get_backend(group) if dist.is_initialized() else "none"


class _Done:
    def wait(self):
        return True


# ---- byte accounting (used by tests and the profile report) ---------------------------------
STATS = {"all_reduce": 0, "reduce_scatter": 0, "all_gather": 0, "all_to_all": 0, "broadcast": 0}


def _count(kind: str, t: torch.Tensor):
    STATS[kind] += t.numel() * t.element_size()


def _prof(kind: str, t: torch.Tensor):
    """roctx range ``comm_issue:<kind>`` + host enqueue time + bytes of one collective (--profile)."""
    from ..utils.profiling import comm_range
    return comm_range(kind, t.numel() * t.element_size())


def _staged(t: torch.Tensor, group) -> bool:
    """gloo with GPU tensors (several ranks rehearsing on one GPU): the transport runs on a host copy,
    synchronously. gloo's own device-tensor path does not complete on this ROCm stack."""
    return t.is_cuda and _backend(group) == "gloo"


def all_reduce(t: torch.Tensor, op=None, group=None, async_op: bool = False):
    if _single(group):
        return _Done()
    _count("all_reduce", t)
    if _use_own(t, group):
        with _prof("all_reduce", t):
            return _own_done(_OWN.all_reduce(t, _own_op(op)), async_op)
    if _staged(t, group):
        h = t.float().cpu() if t.dtype == torch.bfloat16 else t.cpu()
        dist.all_reduce(h, op=op or dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done()
    with _prof("all_reduce", t):
        w = dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=group, async_op=async_op)
    return w if async_op else _Done()


def broadcast(t: torch.Tensor, src: int = 0, group=None, async_op: bool = False):
    if _single(group):
        return _Done()
    _count("broadcast", t)
    if _use_own(t, group):
        with _prof("broadcast", t):
            return _own_done(_OWN.broadcast(t, src), async_op)
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return _Done()
    with _prof("broadcast", t):
        w = dist.broadcast(t, src=src, group=group, async_op=async_op)
    return w if async_op else _Done()


def reduce_scatter(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out (numel n/W) = this rank's slice of the element-wise sum of ``inp`` (numel n)."""
    W = world_size(group)
    if _single(group):
        if out.data_ptr() != inp.data_ptr():
            out.copy_(inp)
        return _Done()
    _count("reduce_scatter", inp)
    if _use_own(inp, group):
        with _prof("reduce_scatter", inp):
            return _own_done(_OWN.reduce_scatter(out, inp, 0), async_op)
    if _backend(group) == "gloo":
        # gloo has no reduce-scatter: sum everything, keep the local slice (tests only)
        tmp = inp.float().cpu() if inp.is_cuda else inp.clone()
        dist.all_reduce(tmp, group=group)
        n = out.numel()
        r = get_rank(group)
        out.copy_(tmp[r * n:(r + 1) * n])
        return _Done()
    with _prof("reduce_scatter", inp):
        w = dist.reduce_scatter_tensor(out, inp, group=group, async_op=async_op)
    return w if async_op else _Done()


def all_gather(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """out (numel n*W) = concatenation of every rank's ``inp`` (numel n). ``inp`` may alias the
    local slice of ``out`` (in-place gather)."""
    W = world_size(group)
    if _single(group):
        if out.data_ptr() != inp.data_ptr():
            out.copy_(inp)
        return _Done()
    _count("all_gather", out)
    if _use_own(out, group):
        with _prof("all_gather", out):
            return _own_done(_OWN.all_gather(out, inp), async_op)
    if _backend(group) == "gloo":
        n = inp.numel()
        src = inp.cpu() if inp.is_cuda else inp.clone()
        parts = [torch.empty_like(src) for _ in range(W)]
        dist.all_gather(parts, src, group=group)
        for r, p in enumerate(parts):
            out[r * n:(r + 1) * n].copy_(p)
        return _Done()
    with _prof("all_gather", out):
        w = dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    return w if async_op else _Done()


def all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None,
               async_op: bool = False):
    """Row-wise all-to-all(v) of 2-D (or 1-D) tensors along dim 0."""
    if _single(group):
        out.copy_(inp)
        return _Done()
    _count("all_to_all", inp)
    if _use_own(inp, group):
        with _prof("all_to_all", inp):
            w = _OWN.all_to_all(out, inp, [int(v) for v in out_splits] if out_splits is not None else [],
                                [int(v) for v in in_splits] if in_splits is not None else [])
            return _own_done(w, async_op)
    if _staged(inp, group):
        # gloo has no all-to-all on device tensors: gather every rank's rows on the host, keep ours
        W, r = world_size(group), get_rank(group)
        src = inp.cpu()
        ins = list(in_splits) if in_splits is not None else [inp.shape[0] // W] * W
        everyone = [None] * W
        dist.all_gather_object(everyone, (src, ins), group=group)
        rows = []
        for peer_rows, peer_splits in everyone:
            start = sum(peer_splits[:r])
            rows.append(peer_rows[start:start + peer_splits[r]])
        out.copy_(torch.cat(rows, 0))
        return _Done()
    with _prof("all_to_all", inp):
        w = dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)
    return w if async_op else _Done()


def barrier(group=None):
    if not _single(group):
        if _OWN is not None and group is None:
            _OWN.barrier()      # through the GPUs: also proves the communicator is alive
        dist.barrier(group=group)
