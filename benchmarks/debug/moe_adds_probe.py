"""Which framework-level elementwise ops a MoE training step still issues (torch profiler, grouped by op and shape)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from distributed_pytorch_amd.config import PRESETS, LLMconfig
from distributed_pytorch_amd.models.llm import LLM

cfg = LLMconfig(**{**LLMconfig().to_dict(), **PRESETS["gpt-moe-8x-swiglu"], "n_layer": 2}).finalize()
model = LLM(cfg).cuda().to(torch.bfloat16)
x = torch.randint(0, cfg.vocab_size, (16, 1024), device="cuda")
def step():
    _, loss, _ = model(x, x)
    loss.backward()
    model.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::mul", "aten::copy_", "aten::sum")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:14s} n={e.count:3d} dev_us={e.device_time_total:9.1f} shapes={e.input_shapes}")
