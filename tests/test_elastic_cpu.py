"""Restart drill: a rank dies mid-run (fault injection), torchrun relaunches the same command line, and with
``--resume auto`` every rank continues from the last periodic resume file to the end of the run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--save_async"]])
def test_ddp_job_survives_a_dead_rank(tmp_path, extra):
    """(--save_async: the generation of iteration 5 is written by a background thread and committed by the loop's poll
    once both ranks' files are complete, a step later: the rank that dies at 7 finds it committed.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", DPA_FAULT_AT_ITER="7", MASTER_ADDR="127.0.0.1",
               GLOO_SOCKET_IFNAME="lo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nproc-per-node", "2",
           "--max-restarts", "1",      # (a c10d rendezvous: every restart gets a fresh store)
           os.path.join(ROOT, "multi-gpu/ddp/train.py"),
           *("--dataset synthetic --synthetic_tokens 65536 --vocab_size 256 --block_size 32 --n_embd 64 --n_layer 2 "
             "--n_head 4 --up_dim 96 --attn mha --pos_emb learn --non_linearity gelu --batch_size 2 "
             "--total_batch_size_str 2*32*2 --max_iters 10 --save_interval 5 --resume auto").split(),
           "--data_dir", str(tmp_path / "data"), "--file_name", str(tmp_path / "run"), *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "[fault injection] rank 1 exits at iteration 7" in out
    assert "resumed from" in out and "at iteration 6" in out, out[-3000:]
    assert any(ln.startswith("step: 10 ") for ln in r.stdout.splitlines()), out[-3000:]
    import json
    m = json.load(open(tmp_path / "run_resume.latest"))
    assert m["iter"] == 10                      # the last generation is committed before the job ends, in both modes


def test_fsdp_job_continues_on_fewer_ranks(tmp_path):
    """A four-rank fully sharded run writes resume files; the same command line started with TWO ranks re-cuts the
    optimizer shards, doubles the gradient accumulation (the global batch is a flag) and trains on; the files that
    the two vanished ranks left of the generation that has aged out are removed by rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", MASTER_ADDR="127.0.0.1", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    flags = ("--dataset synthetic --synthetic_tokens 65536 --vocab_size 256 --block_size 32 --n_embd 64 --n_layer 2 "
             "--n_head 4 --up_dim 96 --attn mha --pos_emb learn --non_linearity gelu --batch_size 2 "
             "--total_batch_size_str 2*32*4 --save_interval 4 --resume auto").split()

    def launch(nproc, max_iters):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nproc-per-node",
               str(nproc), os.path.join(ROOT, "multi-gpu/fsdp/train.py"), *flags, "--max_iters", str(max_iters),
               "--data_dir", str(tmp_path / "data"), "--file_name", str(tmp_path / "run")]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        return r.stdout + r.stderr

    out = launch(4, 9)
    assert any(ln.startswith("step: 9 ") for ln in out.splitlines())
    assert sorted(p.name for p in tmp_path.glob("run_resume.it8.rank*")) == [f"run_resume.it8.rank{r}" for r in range(4)]
    out = launch(2, 14)
    assert "resumed from" in out and "at iteration 9" in out, out[-3000:]
    last = [ln for ln in out.splitlines() if ln.startswith("step: 14 ")]
    assert len(last) == 1 and "grad_accum_steps: 2" in last[0], out[-2000:]
    names = sorted(p.name for p in tmp_path.glob("run_resume.it*"))
    assert names == ["run_resume.it12.rank0", "run_resume.it12.rank1"] + [f"run_resume.it8.rank{r}" for r in range(4)], names
