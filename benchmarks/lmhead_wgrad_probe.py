"""Which accumulation of the lm_head weight gradient over row chunks is fastest:
bf16 GEMM + fp32 add (current), GEMM with fp32 output + add, or addmm straight into the fp32 sum."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

V, C, R = 50304, 768, 8192
lg = torch.randn(R, V, device="cuda", dtype=torch.bfloat16)
xc = torch.randn(R, C, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(V, C, device="cuda", dtype=torch.float32)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def cur():
    ws.add_(torch.mm(lg.t(), xc))


print("bf16 mm + fp32 add      %.1f us" % timeit(cur), flush=True)
try:
    def f32out():
        ws.add_(torch.mm(lg.t(), xc, out_dtype=torch.float32))
    print("fp32-out mm + add       %.1f us" % timeit(f32out), flush=True)
except Exception as e:
    print("fp32-out mm failed:", type(e).__name__, str(e)[:200], flush=True)
try:
    def addmm32():
        torch.addmm(ws, lg.t(), xc, out_dtype=torch.float32, out=ws)
    print("addmm out_dtype fp32    %.1f us" % timeit(addmm32), flush=True)
except Exception as e:
    print("addmm out_dtype failed:", type(e).__name__, str(e)[:200], flush=True)
dwb = torch.zeros(V, C, device="cuda", dtype=torch.bfloat16)
print("bf16 addmm_ (beta=1)    %.1f us" % timeit(lambda: dwb.addmm_(lg.t(), xc)), flush=True)
print("bf16 mm only            %.1f us" % timeit(lambda: torch.mm(lg.t(), xc)), flush=True)
from distributed_pytorch_amd import _ext
ops = _ext.ops()
print("own wgrad2_accum        %.1f us" % timeit(lambda: ops.wgrad2_accum(lg, xc, ws)), flush=True)
