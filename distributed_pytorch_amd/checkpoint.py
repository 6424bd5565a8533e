"""Checkpoints.

``save_reference_checkpoint`` writes the reference's two files
(reference: single-gpu/train.py:361-372):

    <name>_ckpt.pt  = {'model_config', 'train_config', 'model_state'}
    <name>_stats.pt = {'model_config', 'train_config', 'losses', 'total_params', 'active_params'}

``model_state`` uses the un-prefixed key schema of the reference ``LLM`` and fp32 tensors, so
``reference.LLM(cfg).load_state_dict(ckpt['model_state'], strict=True)`` works; the configs are
stored both as dataclass instances (like the reference) and as plain dicts
(``*_config_dict``) that load with ``weights_only=True``.

``save_resume`` / ``load_resume`` add what the reference lacks: optimizer state (per rank for
sharded engines), iteration counter, RNG and data-loader state. A job restarted on ANOTHER number of
ranks reads the files of all ranks of the job that wrote them and re-cuts the optimizer shards
(every engine; the expert-parallel ones collect each routed expert from the old rank that owned it).
"""
from __future__ import annotations

import os
from pathlib import Path

import torch

from .parallel import comm


def save_reference_checkpoint(engine_or_model, model_config, train_config, file_name: str, losses=None,
                              total_params=None, active_params=None, rank: int = 0):
    state = engine_or_model.state_dict()          # collective for sharded engines: call on every rank
    if rank != 0:
        return None
    state = {k: v.detach().to("cpu") for k, v in state.items()}
    ckpt = {"model_config": model_config, "train_config": train_config, "model_state": state,
            "model_config_dict": model_config.to_dict(), "train_config_dict": train_config.to_dict()}
    torch.save(ckpt, file_name + "_ckpt.pt")
    stats = {"model_config": model_config, "train_config": train_config, "losses": losses or {},
             "total_params": total_params, "active_params": active_params,
             "model_config_dict": model_config.to_dict(), "train_config_dict": train_config.to_dict()}
    torch.save(stats, file_name + "_stats.pt")
    return file_name + "_ckpt.pt"


def _marker(path: str) -> str:
    return f"{path}.latest"


def _gen_file(path: str, it: int, rank: int) -> str:
    return f"{path}.it{it}.rank{rank}"


def _read_marker(path: str):
    """(latest iteration, previous iteration or None) from the commit marker, or None without one."""
    import json
    try:
        with open(_marker(path)) as f:
            m = json.load(f)
        return int(m["iter"]), (int(m["prev"]) if m.get("prev") is not None else None)
    except (OSError, ValueError, KeyError):
        return None


def _marker_worlds(path: str) -> dict:
    """{iteration: world size it was written with} for the generations the marker names ({} for markers of
    earlier versions, which did not record the previous generation's)."""
    import json
    try:
        with open(_marker(path)) as f:
            m = json.load(f)
    except (OSError, ValueError):
        return {}
    out = {}
    if m.get("world") is not None and m.get("iter") is not None:
        out[int(m["iter"])] = int(m["world"])
    if m.get("prev") is not None and m.get("prev_world") is not None:
        out[int(m["prev"])] = int(m["prev_world"])
    return out


def _agreed_world(path: str, it, rank: int, device, default: int) -> int:
    """World size of the job that wrote generation ``it``, as RANK 0 reads the marker, on every rank. Collective."""
    w = default
    if rank == 0 and it is not None:
        w = _marker_worlds(path).get(int(it), default)
    t = torch.tensor([w], dtype=torch.int64, device=device)
    comm.broadcast(t, src=0)
    return int(t.item())


def _all_ranks(flag: bool, device) -> bool:
    """True iff ``flag`` holds on every rank."""
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    comm.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return t.item() > 0


_NONE = -1          # encodings of "no such generation" / "no marker" in the broadcast triple
_NO_MARKER = -2


def _agreed_marker(path: str, rank: int, device):
    """The commit marker as RANK 0 reads it, on every rank: (latest, previous | None), or None without one.
    Only rank 0 writes the marker, and with node-local checkpoint directories only its node holds it, so a
    rank that read its own copy could see another state (or none) and derive a different previous generation
    or a different list of resume candidates, i.e. a different number of collectives. Collective."""
    vals = [_NO_MARKER, _NONE]
    if rank == 0:
        m = _read_marker(path)
        if m is not None:
            vals = [m[0], m[1] if m[1] is not None else _NONE]
    t = torch.tensor(vals, dtype=torch.int64, device=device)
    comm.broadcast(t, src=0)
    latest, prev = (int(v) for v in t.tolist())
    if latest == _NO_MARKER:
        return None
    return latest, (prev if prev != _NONE else None)


_PINNED: dict = {}       # staging buffers of the background writer, by position in the state (reused by every save)


def _to_host(obj, where: str = ""):
    """A copy of a (nested) state with every tensor on the host: a snapshot the training loop can no longer change.
    Device tensors go through pinned staging buffers that are allocated once and reused by every later snapshot (a
    pageable copy runs at a tenth of the rate: 1.5 s instead of 0.2 s for 4 GB); the caller makes sure that the
    previous snapshot's writer has finished before it takes the next one."""
    if torch.is_tensor(obj):
        if obj.device.type != "cuda":
            return obj.detach().clone()
        buf = _PINNED.get(where)
        if buf is None or buf.shape != obj.shape or buf.dtype != obj.dtype:
            buf = _PINNED[where] = torch.empty(obj.shape, dtype=obj.dtype, device="cpu", pin_memory=True)
        buf.copy_(obj.detach(), non_blocking=True)
        return buf
    if isinstance(obj, dict):
        return {k: _to_host(v, f"{where}/{k}") for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v, f"{where}/{i}") for i, v in enumerate(obj))
    return obj


class _PendingSave:
    """A generation whose file is being written by a background thread and is not committed yet."""

    def __init__(self, path, it, rank, world, old, old_worlds, thread, error):
        self.path, self.it, self.rank, self.world = path, it, rank, world
        self.old, self.old_worlds, self.thread, self.error = old, old_worlds, thread, error


_PENDING: list = []      # at most one entry: the generation in flight of this process


def finish_pending_save() -> int | None:
    """Wait for the background write of the generation in flight (if any) and commit it: barrier, marker, barrier,
    removal of the generation that aged out. Collective (every rank has the same generation in flight); called by the
    next ``save_resume`` and at the end of training. Returns the committed iteration."""
    if not _PENDING:
        return None
    p = _PENDING.pop()
    p.thread.join()
    if p.error:
        raise RuntimeError(f"writing the resume file of iteration {p.it} failed: {p.error[0]}")
    _commit(p.path, p.it, p.rank, p.world, p.old, p.old_worlds)
    return p.it


def poll_pending_save(device) -> int | None:
    """Commit the generation in flight as soon as EVERY rank's writer has finished: one tiny all-reduce per call while
    a save is pending (nothing otherwise), so the training loop calls it once per step and the new generation becomes
    the resume point within a step of its last file being complete. Collective. Returns the committed iteration."""
    if not _PENDING:
        return None
    done = not _PENDING[0].thread.is_alive()
    if not _all_ranks(done, device):
        return None
    return finish_pending_save()


def _commit(path, it, rank, world, old, old_worlds):
    import json
    comm.barrier()                      # every rank's file of this generation is complete
    prev = old[0] if old is not None and old[0] != it else (old[1] if old is not None else None)
    if rank == 0:
        with open(_marker(path) + ".tmp", "w") as f:
            json.dump({"iter": it, "prev": prev, "world": world, "prev_world": old_worlds.get(prev)}, f)
        os.replace(_marker(path) + ".tmp", _marker(path))
    comm.barrier()                      # the generation is committed
    if old is not None:
        for stale in {old[0], old[1]} - {it, prev, None}:
            # own file; rank 0 also clears what the ranks of a LARGER earlier job left (nobody else would)
            extra = range(world, old_worlds.get(stale, world)) if rank == 0 else ()
            for r in (rank, *extra):
                try:
                    os.remove(_gen_file(path, stale, r))
                except OSError:
                    pass
    # This rank's files of generations the marker never named: written by a job that died before their commit (and
    # whose relaunch did not happen to save at the same iteration again). Nothing can resume from them.
    import glob
    import re
    for f in glob.glob(f"{glob.escape(path)}.it*.rank{rank}"):
        m = re.fullmatch(r"\.it(\d+)\.rank\d+", f[len(path):])
        if m and int(m.group(1)) not in (it, prev):
            try:
                os.remove(f)
            except OSError:
                pass


def save_resume(path: str, engine, it: int, loaders=(), rank: int = 0, world: int = 1, extra=None,
                loader_states=None, background: bool = False):
    """Resume point of iteration ``it`` as one file per rank (optimizer shards are rank-local), committed
    as a set: rank 0 reads the current marker and every rank receives it (before anything changes), every rank
    writes ``<path>.it<it>.rank<r>``, all ranks meet at a barrier, and only then rank 0 atomically replaces the
    marker ``<path>.latest`` = {iter, prev, world}. A job that dies while some ranks are still writing leaves
    the marker on the previous, complete generation, which is kept until the new one is committed; the
    generation before that is deleted afterwards, by every rank from the SAME agreed marker (a rank that
    re-read the file after the barrier could already see the new marker and would keep its old shard forever).

    ``background``: the state is snapshotted to host memory here (a device-to-host copy, the only part the training
    loop waits for), a thread serialises and writes it while training goes on, and the commit (the collectives above)
    happens in ``poll_pending_save()`` (called by the training loop every step: as soon as every rank's file is
    complete), at the latest at the next call or in ``finish_pending_save()``: until then the marker keeps naming the
    previous generation, so a job that dies in between resumes from that one."""
    finish_pending_save()               # at most one generation in flight
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    device = getattr(engine, "device", torch.device("cpu"))
    old = _agreed_marker(path, rank, device)
    old_worlds = _marker_worlds(path) if rank == 0 else {}
    payload = {
        "iter": it,
        "world": world,
        "step_count": getattr(engine, "step_count", None),
        "optimizer": engine.optimizer_state_dict(),
        "buffers": {k: v.detach().cpu() for k, v in engine.module.named_buffers() if v is not None},
        "torch_rng": torch.get_rng_state(),
        "cuda_rng": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
        "loaders": loader_states if loader_states is not None else [l.state_dict() for l in loaders],
        "extra": extra or {},
    }
    mine = _gen_file(path, it, rank)

    def write(state):
        torch.save(state, mine + ".tmp")
        os.replace(mine + ".tmp", mine)

    if background:
        import threading
        snapshot = _to_host(payload)    # (tensors of the live optimizer state would change under the writer)
        if torch.cuda.is_available():
            torch.cuda.synchronize()    # the staged copies are complete before the optimizer runs again
        error: list = []

        def run():
            try:
                write(snapshot)
            except Exception as e:      # noqa: BLE001 - reported by the commit on the main thread
                error.append(repr(e))

        t = threading.Thread(target=run, name=f"resume-writer-it{it}", daemon=False)
        t.start()
        _PENDING.append(_PendingSave(path, it, rank, world, old, old_worlds, t, error))
        return
    write(payload)
    _commit(path, it, rank, world, old, old_worlds)


def resume_candidates(path: str, rank: int = 0, device=None) -> list[int | None]:
    """Iterations to try, newest first: the committed generation, then the one before it; a bare
    ``<path>.rank<r>`` set (the single-generation layout of earlier versions) is tried last as None. With a
    ``device`` the list is rank 0's on every rank (collective); without, this process's own reading."""
    m = _read_marker(path) if device is None else _agreed_marker(path, rank, device)
    out: list[int | None] = []
    if m is not None:
        out.append(m[0])
        if m[1] is not None:
            out.append(m[1])
    out.append(None)
    return out


def find_resume(path: str, rank: int, device) -> int | None | bool:
    """The newest generation whose file exists on EVERY rank: its iteration, None for the legacy
    single-generation layout, or False when there is nothing to resume from. Collective: every rank walks
    rank 0's candidate list, so all of them issue the same sequence of all-reduces."""
    world = comm.world_size()
    for it in resume_candidates(path, rank, device):
        wrote = _agreed_world(path, it, rank, device, world)
        if it is None or wrote == world:
            need = [f"{path}.rank{rank}" if it is None else _gen_file(path, it, rank)]
        else:       # written by a job of another size: this rank reads the files of ALL its ranks
            need = [_gen_file(path, it, r) for r in range(wrote)]
        if _all_ranks(all(os.path.exists(f) for f in need), device):
            return it
    return False


def load_resume(path: str, engine, loaders=(), rank: int = 0, generation: int | None | bool = False) -> int:
    """Load this rank's file of ``generation`` (default: the newest one complete on every rank) and check
    that all ranks resume from the same iteration and optimizer step: ranks restarting from different
    iterations would silently diverge (replicas) or stitch shards of different steps into one model."""
    if generation is False:
        generation = find_resume(path, rank, engine.device)
        if generation is False:
            raise FileNotFoundError(f"no complete resume generation under {path}")
    world = comm.world_size()
    wrote = _agreed_world(path, generation, rank, engine.device, world)
    shards = None
    if generation is not None and wrote != world:
        # A restart on another number of GPUs: every rank reads the files of all ranks of the job that wrote them (they
        # have to be visible to it: one node, or a shared directory) and cuts its own optimizer slice out of them.
        if not getattr(engine, "supports_resharded_resume", False):
            raise RuntimeError(f"the resume files under {path} were written by {wrote} ranks and this job has {world}: "
                               f"the {getattr(engine, 'name', 'this')} engine cannot re-shard them")
        shards = [torch.load(_gen_file(path, generation, r), map_location="cpu", weights_only=False) for r in range(wrote)]
        payload = shards[rank % wrote]
    else:
        fname = f"{path}.rank{rank}" if generation is None else _gen_file(path, generation, rank)
        payload = torch.load(fname, map_location=engine.device, weights_only=False)
    key = torch.tensor([float(payload["iter"]), float(payload.get("step_count") or 0)], dtype=torch.float64,
                       device=engine.device)
    lo, hi = key.clone(), key.clone()
    comm.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    comm.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"resume files under {path} disagree across ranks: iteration / optimizer step "
                           f"{lo.tolist()} .. {hi.tolist()}; refusing to resume from a mixed set")
    if shards is not None:
        engine.load_optimizer_shards([sh["optimizer"] for sh in shards])
        payload = dict(payload, loaders=shards[0]["loaders"])     # the sampling stream is shared by all ranks
        del shards
    else:
        engine.load_optimizer_state_dict(payload["optimizer"])
    with torch.no_grad():
        bufs = dict(engine.module.named_buffers())
        for k, v in payload["buffers"].items():
            if k in bufs and bufs[k] is not None:
                bufs[k].copy_(v.to(bufs[k].device))
    torch.set_rng_state(payload["torch_rng"].cpu())
    if payload["cuda_rng"] is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(payload["cuda_rng"].cpu())
    for l, sd in zip(loaders, payload["loaders"]):
        l.load_state_dict(sd)
    return int(payload["iter"])
