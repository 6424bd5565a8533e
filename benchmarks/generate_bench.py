"""Decode throughput of ``LLM.generate`` (KV cache, one token per step) on one GPU.
usage: python generate_bench.py [preset] [batch] [prompt_len] [new_tokens]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_pytorch_amd.config import PRESETS, LLMconfig
from distributed_pytorch_amd.models.llm import LLM

preset = sys.argv[1] if len(sys.argv) > 1 else "gpt2-124m"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P = int(sys.argv[3]) if len(sys.argv) > 3 else 128
N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
cfg = LLMconfig(**{**LLMconfig().to_dict(), **PRESETS[preset]}).finalize()
torch.manual_seed(0)
model = LLM(cfg).to("cuda").to(torch.bfloat16)
model.eval()
idx = torch.randint(0, cfg.vocab_size, (B, P), device="cuda")
with torch.no_grad():
    for graph in (False, True):
        model.generate(idx, 8, temperature=1.0, topk=50, hip_graph=graph)          # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate(idx, N, temperature=1.0, topk=50, hip_graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert out.shape == (B, P + N) and model.last_decode_graphed == graph
        print(f"{preset} batch {B} prompt {P} {'hipGraph' if graph else 'eager   '}: {N} new tokens in {dt * 1e3:.0f} ms = "
              f"{dt / N * 1e3:.2f} ms/step, {B * N / dt:,.0f} tokens/s", flush=True)
