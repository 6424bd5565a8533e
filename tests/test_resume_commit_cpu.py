"""Resume points are committed as a set: per-rank files of one iteration, a barrier, then one marker.
A rank that died while saving must never make the job resume from a mixed set of iterations."""
import os
import shutil

import pytest
import torch

from dist_utils import run_distributed

from distributed_pytorch_amd.checkpoint import _gen_file, _read_marker, find_resume, load_resume, save_resume
from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

CFG = dict(vocab_size=128, block_size=16, n_embd=32, n_layer=2, n_head=2, up_dim=64, attn="mha", pos_emb="learn",
           non_linearity="gelu")


def _engine(mode):
    torch.manual_seed(0)
    model = LLM(LLMconfig(**CFG).finalize())
    kw = dict(weight_decay=0.1, dtype="fp32", bucket_mb=0.01)
    if mode == "zero2":
        from distributed_pytorch_amd.parallel.zero import ZeroEngine
        return ZeroEngine(model, stage=2, **kw)
    from distributed_pytorch_amd.parallel.ddp import DDPEngine
    return DDPEngine(model, **kw)


def _step(eng, rank, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, CFG["vocab_size"], (4, CFG["block_size"]), generator=g)[rank * 2:(rank + 1) * 2]
    eng.zero_grad()
    _, loss, _ = eng(x, x.roll(-1, dims=1))
    loss.backward()
    eng.clip_grad_norm_(1.0)
    eng.step(1e-2)


def _worker(rank, world, root, mode, scenario):
    path = os.path.join(root, "run_resume")
    eng = _engine(mode)
    for it in range(3):
        _step(eng, rank, it)
        save_resume(path, eng, it, (), rank, world)
    assert _read_marker(path) == (2, 1)
    # generations 2 and 1 are on disk, generation 0 is gone
    assert os.path.exists(_gen_file(path, 2, rank)) and os.path.exists(_gen_file(path, 1, rank))
    assert not os.path.exists(_gen_file(path, 0, rank))
    torch.distributed.barrier()
    if scenario == "lost_file":
        # one rank's file of the committed generation is lost: every rank falls back to the one before
        if rank == 1:
            os.remove(_gen_file(path, 2, rank))
        torch.distributed.barrier()
        assert find_resume(path, rank, eng.device) == 1
        fresh = _engine(mode)
        assert load_resume(path, fresh, (), rank) == 1
        return True
    if scenario == "mixed":
        # one rank's file is replaced by an older iteration's (what independent per-rank replacement used
        # to produce when a rank died mid-save): loading must refuse, not silently diverge
        if rank == 1:
            shutil.copy(_gen_file(path, 1, rank), _gen_file(path, 2, rank))
        torch.distributed.barrier()
        fresh = _engine(mode)
        with pytest.raises(RuntimeError, match="disagree across ranks"):
            load_resume(path, fresh, (), rank)
        return True
    # intact: the newest generation loads and restores the optimizer shards exactly
    fresh = _engine(mode)
    assert load_resume(path, fresh, (), rank) == 2
    for a, b in zip(eng.optims, fresh.optims):
        assert torch.equal(a.master, b.master) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    assert fresh.step_count == eng.step_count
    return True


@pytest.mark.parametrize("scenario", ["intact", "lost_file", "mixed"])
def test_resume_generations_are_committed_as_a_set(tmp_path, scenario):
    assert all(run_distributed(_worker, 2, str(tmp_path), "zero2" if scenario != "mixed" else "ddp", scenario))


def _worker_agreed(rank, world, root, layout):
    """Only rank 0 ever reads the marker; every rank deletes from the value rank 0 saw BEFORE it was replaced,
    whatever the timing between the ranks; with node-local directories (only rank 0's holds the marker) every
    rank still walks the same candidate list."""
    import time

    import distributed_pytorch_amd.checkpoint as ck
    path = os.path.join(root, f"node{rank}" if layout == "node_local" else "shared", "run_resume")
    real_read = ck._read_marker

    def guarded(p):
        assert rank == 0, "a rank other than 0 read the commit marker"
        return real_read(p)

    real_barrier = ck.comm.barrier

    def skewed_barrier(*a, **k):
        # rank 0 runs ahead after every barrier (it replaces the marker at once), rank 1 lags behind it
        real_barrier(*a, **k)
        if rank == 1:
            time.sleep(0.2)

    ck._read_marker = guarded
    ck.comm.barrier = skewed_barrier
    try:
        eng = _engine("zero2")
        for it in range(4):
            _step(eng, rank, it)
            save_resume(path, eng, it, (), rank, world)
            on_disk = sorted(int(f.split(".it")[1].split(".")[0]) for f in os.listdir(os.path.dirname(path))
                             if f".rank{rank}" in f and ".it" in f)
            assert on_disk == [i for i in (it - 1, it) if i >= 0], (rank, it, on_disk)
        torch.distributed.barrier()
        assert find_resume(path, rank, eng.device) == 3
        if rank == 1:
            os.remove(_gen_file(path, 3, rank))
        torch.distributed.barrier()
        assert find_resume(path, rank, eng.device) == 2
        fresh = _engine("zero2")
        assert load_resume(path, fresh, (), rank) == 2
    finally:
        ck._read_marker = real_read
        ck.comm.barrier = real_barrier
    return True


@pytest.mark.parametrize("layout", ["shared", "node_local"])
def test_marker_is_read_by_rank0_only_and_old_generations_go_on_every_rank(tmp_path, layout):
    assert all(run_distributed(_worker_agreed, 2, str(tmp_path), layout))


def _async_worker(rank, world, root, mode):
    """Background writing: the snapshot is the state at the call, the commit comes with the next call (or the final
    flush), and until then the marker names the previous generation."""
    from distributed_pytorch_amd import checkpoint as ck
    path = os.path.join(root, "run_resume")
    eng = _engine(mode)
    _step(eng, rank, 0)
    save_resume(path, eng, 0, (), rank, world, background=True)
    assert len(ck._PENDING) == 1 and _read_marker(path) is None          # written in the background, not committed
    _step(eng, rank, 1)                                                    # training goes on and changes the live state
    truth1 = [o.exp_avg_sq.clone() for o in eng.optims]
    save_resume(path, eng, 1, (), rank, world, background=True)            # commits generation 0, starts generation 1
    torch.distributed.barrier()
    assert _read_marker(path) == (0, None)
    _step(eng, rank, 2)                                                    # the live state moves on past the snapshot
    assert find_resume(path, rank, eng.device) == 0                        # a job dying here resumes from generation 0
    assert ck.finish_pending_save() == 1
    torch.distributed.barrier()
    assert _read_marker(path) == (1, 0) and ck.finish_pending_save() is None
    fresh = _engine(mode)
    assert load_resume(path, fresh, (), rank) == 1
    for o, t in zip(fresh.optims, truth1):
        assert torch.equal(o.exp_avg_sq, t)                                # the snapshot, not the state two steps later
    return True


@pytest.mark.parametrize("mode", ["ddp", "zero2"])
def test_background_save_snapshots_and_commits_late(tmp_path, mode):
    assert all(run_distributed(_async_worker, 2, str(tmp_path), mode))


def _orphan_worker(rank, world, root):
    """A generation written but never committed (its job died before the marker moved) is removed by the next commit."""
    path = os.path.join(root, "run_resume")
    eng = _engine("zero2")
    _step(eng, rank, 0)
    save_resume(path, eng, 0, (), rank, world)
    torch.save({"junk": True}, _gen_file(path, 7, rank))        # what a job that died mid-save leaves behind
    _step(eng, rank, 1)
    save_resume(path, eng, 1, (), rank, world)
    torch.distributed.barrier()
    left = sorted(f for f in os.listdir(root) if f.startswith("run_resume.it") and f.endswith(f".rank{rank}"))
    return left == [f"run_resume.it0.rank{rank}", f"run_resume.it1.rank{rank}"]


def test_uncommitted_generations_are_cleared(tmp_path):
    assert all(run_distributed(_orphan_worker, 2, str(tmp_path)))
