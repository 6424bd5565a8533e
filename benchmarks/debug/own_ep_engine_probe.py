import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
from distributed_pytorch_amd import _ext
from distributed_pytorch_amd.parallel import comm, ep
from distributed_pytorch_amd.config import LLMconfig, Trainconfig
from distributed_pytorch_amd.models.llm import LLM
from distributed_pytorch_amd.trainer import build_engine
torch.cuda.set_device(0)
assert _ext.available()
cls = torch.classes.dpa.RcclComm
own = cls(cls.unique_id(), 0, 1, 0, 0, 0, 8)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29589", rank=0, world_size=1)
comm.FORCE_COLLECTIVES, comm._OWN = True, own
orig = comm.all_to_all
def spy(out, inp, out_splits=None, in_splits=None, group=None, async_op=False):
    r = orig(out, inp, out_splits, in_splits, group, async_op)
    if inp.numel() < 16:
        print("a2a", inp.dtype, tuple(inp.shape), inp.tolist(), "->", out.tolist(), "same storage:", inp.data_ptr() == out.data_ptr(), flush=True)
    return r
comm.all_to_all = spy
cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                pos_emb="learn", non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2).finalize()
tc = Trainconfig(dtype="bf16", learning_rate=1e-3, bucket_mb=0.5)
info = comm.DistInfo(0, 0, 1, torch.device("cuda:0"), "rccl")
torch.manual_seed(1)
eng = build_engine("ep", LLM(cfg), tc, info)
x = torch.randint(0, cfg.vocab_size, (8, 256), device="cuda")
eng.zero_grad()
_, loss, _ = eng(x, x.roll(-1, 1))
loss.backward()
print("loss", loss.item())
