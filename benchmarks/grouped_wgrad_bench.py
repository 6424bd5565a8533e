#!/usr/bin/env python3
"""Per-expert weight gradients of a MoE layer: the 128 x 128 grouped kernel against the grouped mode of the dense
256 x 256 kernel, expert-sorted padded layout with balanced or skewed segments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd import _ext
assert _ext.available()
o = torch.ops.dpa


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


E = 8
for pairs, N, K in ((16384, 3072, 768), (32768, 3072, 768), (65536, 3072, 768), (131072, 3072, 768), (131072, 768, 3072), (524288, 3072, 768)):
    for skew in (0.0, 0.3):
        w = torch.tensor([1.0 + skew * ((i % 3) - 1) for i in range(E)])
        seg = ((w / w.sum() * pairs).long() + 255) // 256 * 256
        rows = int(seg.sum())
        start = torch.cumsum(seg, 0) - seg
        dy = torch.randn(rows, N, device="cuda").bfloat16()
        x = torch.randn(rows, K, device="cuda").bfloat16()
        ss, sr = start.cuda(), seg.cuda()
        flops = 2.0 * rows * N * K
        for on in (True, False):
            o.grouped_wgrad_use_wgrad3(on, True)
            ms = timeit(lambda: o.grouped_wgrad(dy, x, ss, sr))
            print(f"rows {rows:>7} dW[{N},{K}] skew {skew}: {'256x256 grouped' if on else '128x128 grouped'} {ms * 1e3:8.1f} us "
                  f"{flops / ms / 1e9:7.0f} TF", flush=True)
        del dy, x
o.grouped_wgrad_use_wgrad3(True, False)
