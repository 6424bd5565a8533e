"""Allocated / reserved GPU memory over steps for one engine configuration (growth of `allocated` is a leak, growth of
`reserved` alone is allocator fragmentation).   python benchmarks/debug/leak_probe.py block|split"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_pytorch_amd.config import LLMconfig, Trainconfig
from distributed_pytorch_amd.models.llm import LLM
from distributed_pytorch_amd.parallel import comm
from distributed_pytorch_amd.trainer import build_engine

what = sys.argv[1]
cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="gqa", n_kv_heads=2,
                pos_emb="rope", non_linearity="swiglu", act_recomp=(what == "block"), act_recomp_scope="block").finalize()
tc = Trainconfig(learning_rate=1e-3, bucket_mb=48, dtype="bf16", tied_tail="split" if what == "split" else "fused")
if what == "split":
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29977"
    os.environ["DPA_FORCE_DIST"] = "1"
    comm.FORCE_COLLECTIVES = True
    info = comm.init_distributed()
    mode = "ddp"
else:
    info, mode = comm.DistInfo(0, 0, 1, torch.device("cuda:0"), "none"), "single"
torch.manual_seed(0)
eng = build_engine(mode, LLM(cfg), tc, info)
x = torch.randint(0, cfg.vocab_size, (8, 256), device="cuda")
for it in range(1, 301):
    eng.zero_grad()
    _, loss, _ = eng(x, x.roll(-1, 1))
    loss.backward()
    eng.clip_grad_norm_(1.0)
    eng.step(1e-3)
    if it in (5, 50, 100, 200, 300):
        torch.cuda.synchronize()
        print(what, "step", it, "allocated MB %.2f" % (torch.cuda.memory_allocated() / 2**20), "reserved MB %.1f" % (torch.cuda.memory_reserved() / 2**20), flush=True)
if what == "split":
    comm.shutdown()
