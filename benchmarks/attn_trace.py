"""Phase trace of the attention forward (build: python benchmarks/ab_build.py trace attention.hip -DDPA_ATTN_TRACE;
run with DPA_EXT_LIB=benchmarks/ab/_C_trace.so): clocks between the stamps of each key tile, per wave of one workgroup.
Stamps: 0 loop top, 1 after the DMA wait + barrier, 2 after the next tile's DMA issue, 3 after the score MFMAs are
issued (K fragment reads waited for), 4 after the exponentials and row sums (score results waited for), 5 after the
P V MFMAs are issued."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_pytorch_amd._ext import ops as hip
B, T, nh, hs = 64, 1024, 12, 64
q = torch.randn(B, T, nh, hs, device="cuda").bfloat16()
k = torch.randn(B, T, nh, hs, device="cuda").bfloat16()
v = torch.randn(B, T, nh, hs, device="cuda").bfloat16()
for _ in range(3):
    o, lse = hip().attn_fwd(q, k, v, 1 / math.sqrt(hs))
torch.cuda.synchronize()
tr = lse.view(-1)[:448].view(torch.int32).cpu().view(4, 16, 7).long() & 0xFFFFFFFF
names = ["dma wait + barrier", "dma issue", "K reads + score MFMA issue", "exp + sums (score results)", "V reads + PV MFMA issue"]
for w in range(4):
    print(f"wave {w}: clocks per phase and tile (rows: tiles 0..15)")
    for t in range(16):
        row = tr[w, t]
        if row[0] == 0:
            continue
        d = [(int(row[i + 1]) - int(row[i])) & 0xFFFFFFFF for i in range(5)]
        nxt = ((int(tr[w, t + 1, 0]) - int(row[5])) & 0xFFFFFFFF) if t + 1 < 16 and tr[w, t + 1, 0] != 0 else 0
        print(f"  tile {t:2d}: " + "  ".join(f"{n} {x:5d}" for n, x in zip(names, d)) + f"  | to next top {nxt:4d} | total {sum(d) + nxt:5d}")
