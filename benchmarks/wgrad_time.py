"""Median time of the raw weight-gradient kernel (slabs only, no fold) for one shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd import _ext

_ext.require()
ops = _ext.ops()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (65536, 3072, 768)
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
ts = []
for i in range(15):
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    ops.wgrad2_slabs(dy, x)
    b.record()
    b.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
t = ts[len(ts) // 2]
print(f"dW[{N},{K}] M={M}: {t * 1e3:.1f} us  {2.0 * M * N * K / t / 1e9:.0f} TF", flush=True)
