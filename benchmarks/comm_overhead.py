#!/usr/bin/env python3
"""Host cost of issuing one collective: c10d's ProcessGroupNCCL against the framework's own RCCL
communicator, one rank on one GPU (nothing crosses a link, so what is left is the software path:
Python -> dispatcher -> work object -> stream bookkeeping -> ncclAllReduce enqueue -> kernel).

    python benchmarks/comm_overhead.py            # prints a table, one line per (transport, size)

Two numbers per case: host microseconds per call while the GPU keeps up (enqueue cost), and device
microseconds per call from stream events (kernel + launch gaps on the communication stream).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def measure(issue, wait_all, n=200):
    for _ in range(10):
        issue()
    wait_all()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(n):
        issue()
    host = (time.perf_counter() - t0) / n
    wait_all()
    e1.record()
    torch.cuda.synchronize()
    return host * 1e6, e0.elapsed_time(e1) * 1e3 / n


def main():
    from distributed_pytorch_amd import _ext
    assert _ext.available()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    cls = torch.classes.dpa.RcclComm
    own = cls(cls.unique_id(), 0, 1, 0, 0, 0, 0)
    print(f"{'transport':<10} {'op':<15} {'bytes':>12} {'host us/call':>13} {'device us/call':>15}")
    for nbytes in (4, 1 << 20, 48 << 20):
        t = torch.zeros(max(nbytes // 2, 2), dtype=torch.bfloat16, device="cuda")
        out = torch.empty_like(t)
        works = []

        def c10d_ar():
            works.append(dist.all_reduce(t, async_op=True))

        def own_ar():
            works.append(own.all_reduce(t, 0))

        def c10d_rs():
            works.append(dist.reduce_scatter_tensor(out, t, async_op=True))

        def own_rs():
            works.append(own.reduce_scatter(out, t, 0))

        def c10d_ag():
            works.append(dist.all_gather_into_tensor(out, t, async_op=True))

        def own_ag():
            works.append(own.all_gather(out, t))

        def wait_all():
            for w in works:
                w.wait()
            works.clear()

        for name, fn in (("c10d", c10d_ar), ("own", own_ar), ("c10d", c10d_rs), ("own", own_rs), ("c10d", c10d_ag),
                         ("own", own_ag)):
            host, dev = measure(fn, wait_all)
            print(f"{name:<10} {fn.__name__.split('_')[1]:<15} {t.numel() * 2:>12} {host:>13.1f} {dev:>15.1f}", flush=True)
    own.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
