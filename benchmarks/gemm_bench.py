"""Micro-benchmark: weight-gradient GEMM (csrc/gemm_wgrad.hip) against the library GEMM on the
GPT-2 shapes. Interleaved rounds in one process, random operands, median over rounds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd import ops, _ext
_ext.require()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (50304, 768)]:
    Mx = 8192 if N == 50304 else M
    dy = torch.randn(Mx, N, device="cuda").to(torch.bfloat16)
    x = torch.randn(Mx, K, device="cuda").to(torch.bfloat16)
    fl = 2.0 * Mx * N * K
    ref = dy.float().t() @ x.float()
    for st in (1, 4):
        torch.ops.dpa.wgrad2_set_stages(st)
        err = ((ops.weight_grad(dy, x, False, version=2)[0].float() - ref).norm() / ref.norm()).item()
        assert err < 5e-3, (st, err)
    del ref
    t_lib = timeit(lambda: torch.mm(dy.t(), x))
    t_v1 = timeit(lambda: ops.weight_grad(dy, x, False, version=1))
    torch.ops.dpa.wgrad2_set_stages(1)
    t_v2 = timeit(lambda: ops.weight_grad(dy, x, False, version=2))
    torch.ops.dpa.wgrad2_set_stages(4)
    t_v4 = timeit(lambda: ops.weight_grad(dy, x, False, version=2))
    print(f"dW[{N},{K}] M={Mx}: library {t_lib*1e3:8.1f} us ({fl/t_lib/1e9:7.1f} TF)   v1 {t_v1*1e3:8.1f} us ({fl/t_v1/1e9:7.1f} TF)"
          f"   v2 {t_v2*1e3:8.1f} us ({fl/t_v2/1e9:7.1f} TF)   256x256 {t_v4*1e3:8.1f} us ({fl/t_v4/1e9:7.1f} TF)", flush=True)
