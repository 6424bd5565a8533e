"""Restart drill: a rank dies mid-run (fault injection), torchrun relaunches the same command line, and with
``--resume auto`` every rank continues from the last periodic resume file to the end of the run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddp_job_survives_a_dead_rank(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", DPA_FAULT_AT_ITER="7", MASTER_ADDR="127.0.0.1",
               GLOO_SOCKET_IFNAME="lo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nproc-per-node", "2",
           "--max-restarts", "1",      # (a c10d rendezvous: every restart gets a fresh store)
           os.path.join(ROOT, "multi-gpu/ddp/train.py"),
           *("--dataset synthetic --synthetic_tokens 65536 --vocab_size 256 --block_size 32 --n_embd 64 --n_layer 2 "
             "--n_head 4 --up_dim 96 --attn mha --pos_emb learn --non_linearity gelu --batch_size 2 "
             "--total_batch_size_str 2*32*2 --max_iters 10 --save_interval 5 --resume auto").split(),
           "--data_dir", str(tmp_path / "data"), "--file_name", str(tmp_path / "run")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "[fault injection] rank 1 exits at iteration 7" in out
    assert "resumed from" in out and "at iteration 6" in out, out[-3000:]
    assert any(ln.startswith("step: 10 ") for ln in r.stdout.splitlines()), out[-3000:]
