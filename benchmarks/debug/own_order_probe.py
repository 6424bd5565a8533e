"""Does the communication stream really wait for the producer kernels? Out-of-place all_gather / all_to_all
(a real copy on one rank) of a tensor written at the end of a long kernel chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_pytorch_amd import _ext
assert _ext.available()
torch.cuda.set_device(0)
cls = torch.classes.dpa.RcclComm
own = cls(cls.unique_id(), 0, 1, 0, 0, 0, 0)
big = torch.randn(1 << 26, device="cuda")
bad = {"gather": 0, "a2a": 0}
for it in range(30):
    src = torch.zeros(1 << 16, device="cuda")
    dst = torch.full_like(src, -1.0)
    dst2 = torch.full_like(src, -1.0)
    torch.cuda.synchronize()
    for _ in range(10):
        big = big * 1.0000001          # ~ms of queued work
    src += float(it + 1)               # producer: last kernel of the chain
    own.all_gather(dst, src).wait()
    own.all_to_all(dst2, src, [], []).wait()
    ok1 = bool((dst == it + 1).all())
    ok2 = bool((dst2 == it + 1).all())
    bad["gather"] += not ok1
    bad["a2a"] += not ok2
print("stale results out of 30:", bad, flush=True)
