"""Weight gradient of the tied lm_head, dW[V, C] += dlogits[M, V]^T x[M, C] accumulated in fp32 over row chunks: the own
256 x 256 kernel (rows in whole rounds per XCD, the last round cut into row parts: WgradWalk in csrc/host_logic.h)
against the library's fp32 addmm, plus a sweep of the uniform split count on the large layer shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd._ext import ops as hip


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


V, C = 50304, 768
# numerics: two accumulating launches against the fp32 product (half-empty last row of tiles, tiles summed in parts)
for M in (2048, 8192):
    dy = (torch.randn(M, V, device="cuda") * 0.1).bfloat16()
    x = torch.randn(M, C, device="cuda").bfloat16()
    ws = torch.zeros(V, C, device="cuda")
    hip().wgrad2_accum(dy, x, ws)
    hip().wgrad2_accum(dy, x, ws)
    ref = torch.zeros(V, C, device="cuda")
    for r0 in range(0, V, 8192):
        ref[r0:r0 + 8192] = 2 * (dy[:, r0:r0 + 8192].float().t() @ x.float())
    err = ((ws - ref).abs().max() / ref.abs().max()).item()
    print(f"check M={M}: max relative error {err:.2e}", flush=True)
    assert err < 1e-5, err
    del dy, x, ws, ref
for M in (16384, 65536):
    dy = torch.randn(M, V, device="cuda").bfloat16()
    x = torch.randn(M, C, device="cuda").bfloat16()
    ws = torch.zeros(V, C, device="cuda")
    fl = 2.0 * M * V * C
    t_lib = timeit(lambda: torch.addmm(ws, dy.t(), x, out_dtype=torch.float32, out=ws))
    t_own = timeit(lambda: hip().wgrad2_accum(dy, x, ws))
    print(f"head dW[{V},{C}] M={M}: library addmm fp32 {t_lib * 1e3:8.1f} us ({fl / t_lib / 1e9:7.1f} TF)   own {t_own * 1e3:8.1f} us "
          f"({fl / t_own / 1e9:7.1f} TF)", flush=True)
    del dy, x, ws
for M in (16384, 49152):
    for N, K in ((6144, 2048), (8192, 2048), (2048, 8192), (2048, 2048)):
        dy = torch.randn(M, N, device="cuda").bfloat16()
        x = torch.randn(M, K, device="cuda").bfloat16()
        fl = 2.0 * M * N * K
        t_lib = timeit(lambda: torch.mm(dy.t(), x))
        row = f"dW[{N},{K}] M={M}: library {t_lib * 1e3:7.1f} us ({fl / t_lib / 1e9:6.1f} TF) own by splits:"
        for sp in (1, 2, 3, 4, 8):
            if ((N + 255) // 256) * (K // 256) * sp > 2048:
                continue
            hip().wgrad2_force_splits(sp)
            t = timeit(lambda: hip().wgrad2(dy, x))
            row += f"  {sp}: {t * 1e3:7.1f} ({fl / t / 1e9:6.1f})"
        hip().wgrad2_force_splits(0)
        print(row, flush=True)
        del dy, x
