"""Weight-gradient GEMM: hand-written kernels against the library on the layer shapes of the three
GPT-2 sizes (dW[N, K] = dY[M, N]^T X[M, K]); prints which one should serve each shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd import ops, _ext
_ext.require()

def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

CASES = {
    "124m, micro-batch 4": (4096, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
    "124m, micro-batch 8": (8192, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
    "124m, micro-batch 16": (16384, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
    "124m": (65536, [(2304, 768), (768, 768), (3072, 768), (768, 3072)]),
    "350m": (32768, [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]),
    "1.3b": (16384, [(6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192)]),
    "1.3b, micro-batch 48": (49152, [(6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192)]),
}
if os.environ.get("WGRAD_CASES"):
    CASES = {k: v for k, v in CASES.items() if k in os.environ["WGRAD_CASES"].split(";")}
for name, (M, shapes) in CASES.items():
    for (N, K) in shapes:
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        fl = 2.0 * M * N * K
        t_lib = timeit(lambda: torch.mm(dy.t(), x))
        t_own = timeit(lambda: ops.weight_grad(dy, x, False, version=2))
        print(f"{name} dW[{N},{K}] M={M}: library {t_lib*1e3:8.1f} us ({fl/t_lib/1e9:7.1f} TF)   own {t_own*1e3:8.1f} us "
              f"({fl/t_own/1e9:7.1f} TF)   -> {'own' if t_own < t_lib else 'library'}", flush=True)
