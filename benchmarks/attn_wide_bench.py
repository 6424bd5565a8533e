"""Latent attention of the reference's own launch configuration (1024 wide, 8 heads of 128, latent 256, rotary 128) at
the attention level: the per-head form on the (256, 128) flash kernels against the absorbed form on stock SDPA
(q [B, nh, T, 384], one shared key / value head 384 / 256 wide, expanded), forward + backward. FLOPs counted on the
causal half of what each form executes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distributed_pytorch_amd import ops


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for (B, T, nh, dk, dv, lat, rope) in [(16, 1024, 8, 256, 128, 256, 128), (8, 2048, 8, 256, 128, 256, 128), (32, 1024, 12, 128, 64, 96, 64)]:
    q = torch.randn(B, T, nh, dk, device="cuda").bfloat16().requires_grad_()
    k = torch.randn(B, T, nh, dk, device="cuda").bfloat16().requires_grad_()
    v = torch.randn(B, T, nh, dv, device="cuda").bfloat16().requires_grad_()
    g = torch.randn(B, T, nh, dv, device="cuda").bfloat16()
    scale = 1 / math.sqrt(dk)

    def own():
        o = ops.causal_attention(q, k, v, scale)
        o.backward(g)
    t_own = timeit(own)
    fl_own = 3.5 * 2.0 * B * nh * T * T * (dk + dv) / 2        # forward 1x + backward 2.5x of (QK^T + PV)
    # absorbed form: scores latent + rotary wide, values latent wide, one shared head
    wk, wv = lat + rope, lat
    qa = torch.randn(B, nh, T, wk, device="cuda").bfloat16().requires_grad_()
    ka = torch.randn(B, 1, T, wk, device="cuda").bfloat16().requires_grad_()
    va = torch.randn(B, 1, T, wv, device="cuda").bfloat16().requires_grad_()
    ga = torch.randn(B, nh, T, wv, device="cuda").bfloat16()

    def stock():
        o = F.scaled_dot_product_attention(qa, ka.expand(B, nh, T, wk), va.expand(B, nh, T, wv), is_causal=True, scale=1 / math.sqrt(wk))
        o.backward(ga)
    try:
        t_st = timeit(stock, 5)
        st = f"{t_st * 1e3:8.1f} us"
    except Exception as e:   # noqa: BLE001
        t_st, st = None, f"failed ({type(e).__name__})"
    print(f"B{B} T{T} nh{nh}: per-head ({dk},{dv}) flash fwd+bwd {t_own * 1e3:8.1f} us ({fl_own / t_own / 1e9:6.1f} TF)   "
          f"absorbed ({wk},{wv}) stock SDPA fwd+bwd {st}" + (f"   -> {t_st / t_own:.1f}x" if t_st else ""), flush=True)
