#!/usr/bin/env python3
"""Per-shape time of the decode-step linear kernel (csrc/decode.hip) against the stock chain it replaces,
both replayed from a hipGraph of 12 dependent calls (what a decode step looks like to the GPU).

    python benchmarks/decode_linear_bench.py [batch] [n_embd] [up_dim] [vocab]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_pytorch_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 768
U = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
V = int(sys.argv[4]) if len(sys.argv) > 4 else 50304
dev = "cuda"


def graph_time(fn, reps=12, iters=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


def case(name, K, N, ln, bias, res, act):
    x = torch.randn(B, K, device=dev).bfloat16()
    w = (torch.randn(N * (2 if act == ops.ACT_SWIGLU else 1), K, device=dev) / K ** 0.5).bfloat16()
    lw, lb = torch.ones(K, device=dev).bfloat16(), torch.zeros(K, device=dev).bfloat16()
    b = torch.zeros(w.shape[0], device=dev).bfloat16() if bias else None
    r = torch.randn(B, N, device=dev).bfloat16() if res else None
    lnp = (lw, lb, 1e-5) if ln else None
    state = {"x": x}

    def fused():
        state["y"] = ops.decode_linear(state["x"], lnp, w, b, r, act)

    def stock():
        h = state["x"]
        if ln:
            h = ops.layer_norm(h, lw, lb, 1e-5)
        y = F.linear(h, w, b)
        if act == ops.ACT_GELU:
            y = ops.gelu(y)
        elif act == ops.ACT_SWIGLU:
            y = ops.swiglu(y)
        if res:
            y = y + r
        state["y"] = y

    tf, ts = graph_time(fused), graph_time(stock)
    gbs = w.numel() * 2 / (tf * 1e-6) / 1e9
    print(f"  {name:34s} K {K:5d} N {N:6d}: fused {tf:7.2f} us ({gbs:6.0f} GB/s of weights) | stock chain {ts:7.2f} us")


print(f"decode_linear, batch {B}")
case("LN + qkv + bias", C, 3 * C, True, True, False, ops.ACT_NONE)
case("attention projection + bias + res", C, C, False, True, True, ops.ACT_NONE)
case("LN + c_fc + GELU", C, U, True, False, False, ops.ACT_GELU)
case("LN + c_fc + SwiGLU", C, U, True, False, False, ops.ACT_SWIGLU)
case("c_proj + res", U, C, False, False, True, ops.ACT_NONE)
case("LN + head", C, V, True, False, False, ops.ACT_NONE)
