"""Spawn helpers for multi-process CPU (gloo) tests."""
import os
import socket
import traceback

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry(rank, world, port, fn, args, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = fn(rank, world, *args)
        ret[rank] = ("ok", out)
    except Exception:  # pragma: no cover - surfaced by the parent
        ret[rank] = ("err", traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def run_distributed(fn, world: int, *args):
    """Run ``fn(rank, world, *args)`` on ``world`` gloo ranks; returns the list of results."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    mp.spawn(_entry, args=(world, port, fn, args, ret), nprocs=world, join=True)
    outs = []
    for r in range(world):
        status, val = ret[r]
        assert status == "ok", f"rank {r} failed:\n{val}"
        outs.append(val)
    return outs
