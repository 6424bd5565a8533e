import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from distributed_pytorch_amd import _ext, ops
from distributed_pytorch_amd.parallel import comm
assert _ext.available()
torch.cuda.set_device(0)
cls = torch.classes.dpa.RcclComm
own = cls(cls.unique_id(), 0, 1, 0, 0, 0, 0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29589", rank=0, world_size=1)
comm.FORCE_COLLECTIVES, comm._OWN = True, own
torch.manual_seed(0)
idx = torch.randint(0, 3, (1024, 2), device="cuda")
row_of_pair, counts = ops.plan_dense(idx, 3)
print("counts", counts.dtype, counts.shape, counts.stride(), counts.storage_offset(), counts.is_contiguous(), counts.tolist())
recv = torch.full_like(counts, -7)
comm.all_to_all(recv, counts)
print("recv after comm.all_to_all", recv.tolist())
recv2 = torch.full_like(counts, -7)
own.all_to_all(recv2, counts, [], []).wait()
print("recv after own.all_to_all", recv2.tolist())
