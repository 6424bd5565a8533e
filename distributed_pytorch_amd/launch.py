"""Notebook entry: one command starts the whole single-node job.

The reference's ``kaggle-*.py`` scripts are what a notebook cell runs (reference: multi-gpu/ddp/kaggle-ddp.py:5,
``!torchrun --standalone --nproc_per_node=2 kaggle-train.py ...``). Here the same scripts also run as a plain
``!python kaggle-ddp.py ...``: started outside a launcher, ``self_launch`` counts the visible GPUs, starts
``torch.distributed.run`` as a CHILD process with one rank per GPU on the loopback rendezvous, streams its output
and exits with its code. The parent never touches the GPU (``torch.cuda.device_count()`` does not initialise it),
so nothing that holds a device is ever replaced by another program. Under a launcher (RANK / WORLD_SIZE in the
environment) the call returns at once and the script goes on as that rank.

    python kaggle-ddp.py --nproc 4 --max_restarts 2 <training flags>      # 4 ranks, relaunch up to twice on a failure
    DPA_NPROC=2 python kaggle-fsdp.py <training flags>
"""
from __future__ import annotations

import os
import subprocess
import sys


def _pop_flag(argv: list[str], name: str) -> str | None:
    """Remove ``--name V`` / ``--name=V`` from argv; returns V (None if absent)."""
    for i, a in enumerate(argv):
        if a == f"--{name}" and i + 1 < len(argv):
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        if a.startswith(f"--{name}="):
            del argv[i]
            return a.split("=", 1)[1]
    return None


def launcher_command(script: str, argv: list[str], nproc: int, max_restarts: int = 0, port: int = 0) -> list[str]:
    """The ``torch.distributed.run`` line for ``nproc`` ranks of ``script`` on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           f"--max-restarts={max_restarts}"]
    if port > 0:
        cmd += ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    else:
        cmd += ["--standalone", "--local-addr", "127.0.0.1"]   # the container's hostname may not resolve
    return cmd + [script, *argv]


def self_launch(script: str | None = None, argv: list[str] | None = None) -> None:
    """Called first thing by a notebook entry script. Returns if this process is already a rank of a launched job
    or the job has one rank; otherwise runs the launcher as a child and exits with its return code."""
    argv = sys.argv[1:] if argv is None else argv
    nproc = _pop_flag(argv, "nproc")
    restarts = _pop_flag(argv, "max_restarts")
    port = _pop_flag(argv, "master_port")
    if argv is not sys.argv:
        sys.argv[1:] = argv
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ or "TORCHELASTIC_RUN_ID" in os.environ:
        return
    n = int(nproc or os.environ.get("DPA_NPROC", "0") or 0)
    if n <= 0:
        import torch
        n = torch.cuda.device_count()
    if n <= 1:
        return
    script = script or os.path.abspath(sys.argv[0])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC between the ranks' devices
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = launcher_command(script, argv, n, int(restarts or 0), int(port or 0))
    print(f"[launch] {n} ranks: {' '.join(cmd)}", flush=True)
    sys.exit(subprocess.call(cmd, env=env))
