"""Graphed decode of one preset under rocprofv3: python decode_profile.py [preset] [batch] [new_tokens].
Only the graphed generate() runs (one warm-up call, one measured call), so the kernel statistics of the
trace are dominated by decode steps: calls / (2 * new_tokens) is the per-step count of a kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd.config import PRESETS, LLMconfig
from distributed_pytorch_amd.models.llm import LLM

preset = sys.argv[1] if len(sys.argv) > 1 else "gpt2-124m"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cfg = LLMconfig(**{**LLMconfig().to_dict(), **PRESETS[preset]}).finalize()
torch.manual_seed(0)
model = LLM(cfg).to("cuda").to(torch.bfloat16).eval()
idx = torch.randint(0, cfg.vocab_size, (B, 16), device="cuda")
with torch.no_grad():
    model.generate(idx, N, temperature=1.0, topk=50, hip_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.generate(idx, N, temperature=1.0, topk=50, hip_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{preset} batch {B}: {N} new tokens in {dt * 1e3:.0f} ms = {dt / N * 1e3:.3f} ms/step (prompt + capture included)")
