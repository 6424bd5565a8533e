"""The two library products of the head's backward (dx = dlogits W, dW += dlogits^T x with an fp32 sum) with and
without PyTorch's TunableOp picking the library solution: python lmhead_tunable_probe.py [rows]"""
import os
import sys
import time

import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
V, C = 50304, 768
dev = "cuda"
lg = (torch.randn(rows, V, device=dev) * 0.01).bfloat16()
w = (torch.randn(V, C, device=dev) * 0.02).bfloat16()
xc = torch.randn(rows, C, device=dev).bfloat16()
dx = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
ws = torch.zeros(V, C, device=dev, dtype=torch.float32)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def f_dx():
    torch.mm(lg, w, out=dx)


def f_dw():
    torch.addmm(ws, lg.t(), xc, out_dtype=torch.float32, out=ws)


fl = 2.0 * rows * V * C
for label in ("default", "tunable"):
    if label == "tunable":
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(3000)
        torch.cuda.tunable.set_max_tuning_iterations(20)
        torch.cuda.tunable.set_filename(os.environ.get("TUNABLE_OUT", "/tmp/tunableop_lmhead.csv"))
        t0 = time.time()
        f_dx()
        f_dw()
        torch.cuda.synchronize()
        print(f"tuning took {time.time() - t0:.1f} s", flush=True)
    tdx, tdw = timeit(f_dx), timeit(f_dw)
    print(f"{label:8s} rows {rows}: dx {tdx:6.3f} ms ({fl / tdx / 1e9:6.0f} TF)   dW {tdw:6.3f} ms ({fl / tdw / 1e9:6.0f} TF)", flush=True)
# measured on MI355X at 65536 rows: dx 4.03 -> 3.90 ms, dW 5.32 -> 5.30 ms after 164 s of tuning: not worth a tuned
# results file in the training path
