"""Time the fused cross-entropy kernel on one logits chunk of the head ([65536, 50304] bf16, in place) for the
build named by DPA_EXT_LIB (benchmarks/ab_build.py xntN xent.hip -DDPA_XENT_NT=N: bit 0 non-temporal loads, bit 1
non-temporal stores)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from distributed_pytorch_amd._ext import ops

rows, V = 65536, 50304
torch.manual_seed(0)
base = (torch.randn(rows // 8, V, device="cuda") * 2).to(torch.bfloat16).repeat(8, 1)
lg = torch.empty_like(base)
tg = torch.randint(0, V, (rows,), device="cuda")
gs = torch.full((1,), 1.0 / rows, device="cuda")
ts = []
for i in range(12):
    lg.copy_(base)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loss = ops().xent_fused(lg, tg, gs, -1, True)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
byt = 2 * rows * V * 2
print(f"{os.environ.get('DPA_EXT_LIB', 'default')}: median {ts[len(ts) // 2] * 1e3:.0f} us  min {ts[0] * 1e3:.0f} us  "
      f"{byt / ts[len(ts) // 2] / 1e9:.2f} TB/s  loss {float(loss.sum() * gs[0]):.4f}")
