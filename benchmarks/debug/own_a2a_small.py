import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_pytorch_amd import _ext
assert _ext.available()
torch.cuda.set_device(0)
cls = torch.classes.dpa.RcclComm
own = cls(cls.unique_id(), 0, 1, 0, 0, 0, 0)
for n in (1, 2, 3, 4, 5, 8, 16):
    for dt in (torch.int32, torch.int64, torch.float32):
        c = (torch.arange(n, device="cuda") + 5).to(dt)
        r = torch.full_like(c, -7)
        own.all_to_all(r, c, [], []).wait()
        print(n, dt, r.tolist(), flush=True)
