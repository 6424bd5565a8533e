"""Next-token sampling: the one-kernel sampler against the chain of stock ops (V = 50304, top-k 50)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd import ops


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B in (8, 64):
    logits = torch.randn(B, 50304, device="cuda").to(torch.bfloat16)
    seed = torch.tensor([1], dtype=torch.int64, device="cuda")
    print(f"B={B}: fused kernel {timeit(lambda: ops.sample_fused(logits, 0.8, 50, seed)):7.1f} us   "
          f"fused + seed draw {timeit(lambda: ops.sample_fused(logits, 0.8, 50)):7.1f} us   "
          f"stock chain {timeit(lambda: ops.sample_eager(logits, 0.8, 50)):7.1f} us", flush=True)
