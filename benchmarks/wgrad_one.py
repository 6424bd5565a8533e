"""One weight-gradient shape, a few calls: the target of a counter-collection run.
usage: python wgrad_one.py [M N K [calls]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd import _ext, ops

_ext.require()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (65536, 3072, 768)
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
for _ in range(calls):
    ops.weight_grad(dy, x, False, version=2)
torch.cuda.synchronize()
print("done", flush=True)
