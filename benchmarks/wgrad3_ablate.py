"""wgrad3_kernel on the GPT-2-124M layer shapes at the micro-batch of the headline run (M = 262144) and at 65536:
time and TF/s of the loaded build (run once per ablation build, see DPA_WG3_ABLATE in csrc/gemm_wgrad.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd._ext import ops as hip


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


print("build:", os.environ.get("DPA_EXT_LIB", "default"))
dy = torch.randn(8192, 768, device="cuda").bfloat16()
x = torch.randn(8192, 1024, device="cuda").bfloat16()
ref = dy.float().t() @ x.float()
got = hip().wgrad2(dy, x).float()
print(f"check: max relative error {((got - ref).abs().max() / ref.abs().max()).item():.2e}")
_, db = hip().wgrad2_slabs_bias(dy, x)
refb = dy.float().sum(0)
print(f"check: bias gradient max relative error {((db.float() - refb).abs().max() / refb.abs().max()).item():.2e}")
for M in (65536, 262144):
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, K, device="cuda").bfloat16()
        t = timeit(lambda: hip().wgrad2(dy, x))
        tb = timeit(lambda: hip().wgrad2_slabs_bias(dy, x))
        print(f"M={M:6d} dW[{N:4d},{K:4d}]  {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF   with the bias gradient (slabs, no fold) {tb * 1e3:8.1f} us",
              flush=True)
        del dy, x
