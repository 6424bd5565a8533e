"""Shader clock and socket power while one kernel runs back to back (rocm-smi sampled from a thread):
shows how far the sustained clock of a kernel is from the 2.4 GHz that peak figures assume.

    python benchmarks/clock_under_load.py [wgrad|gemm|attn|idle] [seconds]
"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd._ext import ops as hip

what = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
M = 262144
if what == "wgrad":
    dy, x = torch.randn(M, 3072, device="cuda").bfloat16(), torch.randn(M, 768, device="cuda").bfloat16()
    fn = lambda: hip().wgrad2(dy, x)
    flops = 2.0 * M * 3072 * 768
elif what == "gemm":
    a, w = torch.randn(M, 768, device="cuda").bfloat16(), torch.randn(2304, 768, device="cuda").bfloat16()
    fn = lambda: hip().gemm_nt(a, w, None, 0, None)
    flops = 2.0 * M * 2304 * 768
elif what == "attn":
    q = torch.randn(64, 1024, 12, 64, device="cuda").bfloat16()
    fn = lambda: hip().attn_fwd(q, q, q, 0.125)
    flops = 4.0 * 64 * 12 * 1024 * 1024 * 64 / 2
else:
    fn, flops = (lambda: time.sleep(0.01)), 0.0

samples = []
stop = False


def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        keep = [" ".join(ln.split()) for ln in r.stdout.splitlines() if "sclk" in ln or "(W)" in ln]
        samples.append(" | ".join(keep))
        time.sleep(0.5)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n += 50
el = time.time() - t0
stop = True
th.join()
print(f"{what}: {n} calls in {el:.2f} s, {el / n * 1e6:.1f} us per call, {flops * n / el / 1e12:.0f} TF/s")
for s in samples[:: max(1, len(samples) // 6)]:
    print("  ", s[-300:])
