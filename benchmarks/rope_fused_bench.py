#!/usr/bin/env python3
"""Rotary embedding inside the qkv GEMM's epilogue / the attention backward's stores against separate passes.

    python benchmarks/rope_fused_bench.py [--batch 64] [--seq 1024] [--nh 12] [--nkv 4] [--hs 64]

Prints the time of the projection alone (plain, plain + in-place rotary pass, rotary epilogue) and of the
whole block front end (projection + rotary + attention, forward + backward) both ways.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_pytorch_amd import _ext, ops  # noqa: E402
from distributed_pytorch_amd.ops.rope import rope_tables  # noqa: E402


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return t[len(t) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--nh", type=int, default=12)
    ap.add_argument("--nkv", type=int, default=4)
    ap.add_argument("--hs", type=int, default=64)
    a = ap.parse_args()
    o = _ext.require()
    B, T, nh, nkv, hs = a.batch, a.seq, a.nh, a.nkv, a.hs
    C, W = nh * hs, (nh + 2 * nkv) * hs
    cos_t, sin_t = rope_tables(T, hs, device="cuda")
    x = torch.randn(B, T, C, device="cuda").bfloat16().requires_grad_()
    w = (torch.randn(W, C, device="cuda") / math.sqrt(C)).bfloat16().requires_grad_()
    b = torch.zeros(W, device="cuda").bfloat16().requires_grad_()
    x2 = x.detach().view(-1, C)
    do = torch.randn(B, T, nh, hs, device="cuda").bfloat16()
    print(f"B {B} T {T} nh {nh} nkv {nkv} hs {hs}: projection [{B * T}, {C}] x [{W}, {C}]^T")
    t_plain = timeit(lambda: o.gemm_nt(x2, w.detach(), b.detach(), 0, None))
    buf = o.gemm_nt(x2, w.detach(), b.detach(), 0, None)[0].view(B, T, W)
    t_pass = timeit(lambda: o.rope_(buf[..., : (nh + nkv) * hs].view(B, T, nh + nkv, hs), cos_t, sin_t, 0, False))
    t_epi = timeit(lambda: o.gemm_nt_rope(x2, w.detach(), b.detach(), cos_t, sin_t, 0, T, (nh + nkv) * hs, hs))
    print(f"  projection plain {t_plain:8.1f} us | + rotary pass {t_pass:7.1f} us = {t_plain + t_pass:8.1f} us | rotary epilogue {t_epi:8.1f} us")

    def block(fused):
        for t in (x, w, b):
            t.grad = None
        if fused:
            out = ops.qkv_rope_attention(x, w, b, cos_t, sin_t, 0, nh, nkv)
        else:
            qkv = ops.rope_qkv_(ops.linear(x, w, b), cos_t, sin_t, 0, nh + nkv, hs)
            out = ops.causal_self_attention_qkv(qkv, nh, nkv)
        out.backward(do)

    t_sep = timeit(lambda: block(False), n=10, warm=3)
    t_fused = timeit(lambda: block(True), n=10, warm=3)
    print(f"  projection + rotary + attention, forward + backward: separate passes {t_sep:9.1f} us | fused {t_fused:9.1f} us")


if __name__ == "__main__":
    main()
