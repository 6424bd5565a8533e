"""Micro-benchmark of the flash-attention kernels (causal, bf16, random data): TFLOP/s of the
forward and of the backward (dK/dV + dQ kernels), counting the causal half of the FLOPs."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
NIT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
from distributed_pytorch_amd._ext import ops as hip

def timeit(fn, n=None):
    n = n or NIT
    for _ in range(min(3, n)): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

CASES = [(32, 1024, 12, 12, 64), (16, 1024, 16, 16, 128), (8, 4096, 12, 12, 64), (16, 2048, 32, 8, 128),
         (64, 1024, 12, 12, 64)]
if len(sys.argv) > 1:                      # python attn_bench.py <case index> [iterations]
    CASES = [CASES[int(sys.argv[1])]]
for (B, T, nh, nkv, hs) in CASES:
    q = torch.randn(B, T, nh, hs, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, T, nkv, hs, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, T, nkv, hs, device="cuda").to(torch.bfloat16)
    scale = 1 / math.sqrt(hs)
    o, lse = hip().attn_fwd(q, k, v, scale)
    do = torch.randn_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    fl = 4.0 * B * nh * T * T * hs / 2
    tf = timeit(lambda: hip().attn_fwd(q, k, v, scale))
    tb = timeit(lambda: hip().attn_bwd(do, q, k, v, o, lse, dq, dk, dv, scale))
    print(f"B{B} T{T} nh{nh} nkv{nkv} hs{hs}: fwd {tf*1e3:7.1f} us ({fl/tf/1e9:6.1f} TF)  bwd {tb*1e3:7.1f} us ({2.5*fl/tb/1e9:6.1f} TF on 5 products; 7 executed)", flush=True)
