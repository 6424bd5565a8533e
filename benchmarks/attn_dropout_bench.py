"""Cost of in-kernel attention dropout: forward / backward time of the flash kernels with p = 0 and 0.1."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd._ext import ops as hip

B, T, nh, hs = 64, 1024, 12, 64
q, k, v = (torch.randn(B, T, nh, hs, device="cuda").to(torch.bfloat16) for _ in range(3))
scale = 1 / math.sqrt(hs)
seed = torch.tensor([42], dtype=torch.int64, device="cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for p in (0.0, 0.1):
    sd = seed if p > 0 else None
    o, lse = hip().attn_fwd(q, k, v, scale, p, sd)
    do = torch.randn_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    tf = timeit(lambda: hip().attn_fwd(q, k, v, scale, p, sd))
    tb = timeit(lambda: hip().attn_bwd(do, q, k, v, o, lse, dq, dk, dv, scale, p, sd))
    print(f"B{B} T{T} nh{nh} hs{hs} dropout {p}: fwd {tf:7.1f} us  bwd {tb:7.1f} us", flush=True)
