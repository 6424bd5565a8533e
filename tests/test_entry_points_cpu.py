"""Every training entry script (the reference's file layout) starts, trains a few steps on the CPU and exits
cleanly as a single process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ("--dataset synthetic --synthetic_tokens 65536 --vocab_size 256 --block_size 32 --n_embd 64 --n_layer 2 "
          "--n_head 4 --up_dim 96 --attn mha --pos_emb learn --non_linearity gelu --batch_size 2 "
          "--total_batch_size_str 2*32*2 --max_iters 3").split()
SCRIPTS = ["single-gpu/train.py", "multi-gpu/ddp/train.py", "multi-gpu/ddp/kaggle-ddp.py", "multi-gpu/fsdp/train.py",
           "multi-gpu/fsdp/kaggle-fsdp.py", "multi-gpu/zero1,2/train_zero1.py", "multi-gpu/zero1,2/train_zero2.py",
           "multi-gpu/zero1,2/kaggle-zero1.py", "multi-gpu/zero1,2/kaggle-zero2.py", "multi-gpu/ep/train_ep.py",
           "multi-gpu/ep/train_ep_fsdp.py"]


@pytest.mark.parametrize("script", SCRIPTS)
def test_entry_script_trains(script, tmp_path):
    extra = "--moe --n_exp 4 --n_shared 0 --n_act 2".split() if "/ep/" in script else []
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    cmd = [sys.executable, os.path.join(ROOT, script), *COMMON, *extra, "--data_dir", str(tmp_path / "data"),
           "--file_name", str(tmp_path / "out")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    steps = [ln for ln in r.stdout.splitlines() if ln.startswith("step: 3 ")]
    assert len(steps) == 1 and "train loss" in steps[0], r.stdout[-1500:]


def test_sample_script_reads_saved_model(tmp_path):
    """--save_model writes a checkpoint that single-gpu/sample.py loads and samples from, with no model flags."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    common = [a if a != "256" else "257" for a in COMMON]          # the byte tokeniser's vocabulary
    cmd = [sys.executable, os.path.join(ROOT, "single-gpu/train.py"), *common, "--data_dir", str(tmp_path / "data"),
           "--file_name", str(tmp_path / "out"), "--save_model"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = [sys.executable, os.path.join(ROOT, "single-gpu/sample.py"), "--ckpt", str(tmp_path / "out_ckpt.pt"), "--prompt", "To be",
           "--max_new_tokens", "40", "--num_samples", "2", "--tokenizer", "bytes", "--token_ids", "--device", "cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("[") and not ln.startswith("[sample]")]
    assert len(rows) == 2 and all(len(x) == 5 + 40 for x in rows), r.stdout[-800:]
    assert all(x[:5] == list(b"To be") and max(x) < 257 for x in rows)


@pytest.mark.parametrize("script", ["multi-gpu/ddp/kaggle-ddp.py", "multi-gpu/zero1,2/kaggle-zero2.py"])
def test_notebook_script_launches_its_own_ranks(script, tmp_path):
    """Started as a plain process with --nproc 2, a notebook entry script runs the launcher as a child: two gloo
    ranks train, rank 0's log lines come back through the parent, and the parent exits with the job's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    env["OMP_NUM_THREADS"] = "2"
    common = [a if a != "2*32*2" else "2*32*2*2" for a in COMMON]   # two ranks x two micro-steps
    cmd = [sys.executable, os.path.join(ROOT, script), "--nproc", "2", *common, "--data_dir", str(tmp_path / "data"),
           "--file_name", str(tmp_path / "out")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "[launch] 2 ranks:" in r.stdout
    steps = [ln for ln in r.stdout.splitlines() if ln.startswith("step: 3 ")]
    assert len(steps) == 1 and "train loss" in steps[0], r.stdout[-1500:]


def test_launcher_flags_are_stripped_and_a_launched_rank_returns(monkeypatch):
    from distributed_pytorch_amd import launch
    argv = ["--nproc", "4", "--moe", "--max_restarts=2", "--n_exp", "8"]
    monkeypatch.setenv("RANK", "0")
    launch.self_launch("x.py", argv)            # a launched rank: returns, launcher flags removed
    assert argv == ["--moe", "--n_exp", "8"]
    cmd = launch.launcher_command("x.py", argv, 4, max_restarts=2)
    assert "--nproc-per-node=4" in cmd and "--max-restarts=2" in cmd and "--standalone" in cmd
    assert cmd[-4:] == ["x.py", "--moe", "--n_exp", "8"]
    cmd = launch.launcher_command("x.py", argv, 2, port=29511)
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--standalone" not in cmd
