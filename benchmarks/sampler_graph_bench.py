"""Kernel-side time of the one-kernel sampler replayed from a hipGraph (12 calls per replay), by top-k setting:
0 = no filter, 50 = search over the candidates above the bound, 600 = search over the whole row."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
logits = (torch.randn(B, 50304, device="cuda") * 2.0).to(torch.bfloat16)
seed = torch.tensor([1], dtype=torch.int64, device="cuda")
for topk in (0, 1, 50, 512, 600):
    fn = lambda: ops.sample_fused(logits, 0.8, topk, seed)
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(12):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        g.replay()
    b.record()
    b.synchronize()
    print(f"B={B} top-k {topk:4d}: {a.elapsed_time(b) / 600 * 1e3:7.1f} us per call", flush=True)
