#!/usr/bin/env python3
"""lm_head + cross-entropy pieces at GPT-2 shape (C = 768, V = 50304), per row-chunk size: the logits GEMM,
the cross-entropy kernel, dx = dlogits @ W and dW += dlogits^T @ x, library against own kernels.

    python benchmarks/lmhead_bench.py            # table: ms per chunk and ms per 65536 rows
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from distributed_pytorch_amd import _ext  # noqa: E402


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


def main():
    assert _ext.available()
    o = torch.ops.dpa
    C, V, N = 768, 50304, 65536
    torch.manual_seed(0)
    w = (torch.randn(V, C, device="cuda") * 0.02).bfloat16()
    wt = w.t().contiguous()
    print(f"{'rows':>6} {'piece':<28} {'ms/chunk':>9} {'ms/65536 rows':>14} {'TF/s':>7}")
    for R in (2048, 4096, 8192, 16384):
        x = torch.randn(R, C, device="cuda").bfloat16()
        tg = torch.randint(0, V, (R,), device="cuda")
        lg = torch.empty(R, V, device="cuda", dtype=torch.bfloat16)
        dx = torch.empty(R, C, device="cuda", dtype=torch.bfloat16)
        ws = torch.zeros(V, C, device="cuda", dtype=torch.float32)
        gs = torch.full((1,), 1.0 / R, device="cuda")
        flops = 2.0 * R * C * V
        cases = [
            ("logits  library mm", lambda: torch.mm(x, w.t(), out=lg), flops),
            ("logits  own gemm_nt", lambda: o.gemm_nt(x, w, None, 0, None), flops),
            ("xent_fused (in place)", lambda: o.xent_fused(lg, tg, gs, -1, True), 0.0),
            ("dx      library mm", lambda: torch.mm(lg, w, out=dx), flops),
            ("dx      own gemm_nt(W^T)", lambda: o.gemm_nt(lg, wt, None, 0, None), flops),
            ("dW      library addmm fp32", lambda: torch.addmm(ws, lg.t(), x, out_dtype=torch.float32, out=ws), flops),
            ("dW      own wgrad2_accum", lambda: o.wgrad2_accum(lg, x, ws), flops),
        ]
        for name, fn, fl in cases:
            try:
                ms = timeit(fn)
            except Exception as e:   # noqa: BLE001 - report unsupported shapes in the table
                print(f"{R:>6} {name:<28} unsupported: {str(e)[:60]}")
                continue
            tf = fl / (ms * 1e-3) / 1e12 if fl else 0.0
            print(f"{R:>6} {name:<28} {ms:>9.3f} {ms * N / R:>14.2f} {tf:>7.0f}", flush=True)
        del x, lg, dx, ws


if __name__ == "__main__":
    main()
