"""Throughput of the REFERENCE model + training loop on this GPU, for the baseline row.

Usage:  python benchmarks/reference_baseline.py /path/to/reference/single-gpu/model.py [batch]

Imports the reference ``model.py`` by path (source only) and runs its single-GPU training recipe
unchanged in spirit (reference: single-gpu/train.py:17-25,314-352): fp32 parameters, bf16
autocast, ``F.scaled_dot_product_attention``, ``torch.optim.AdamW(fused=True)``, global-norm clip,
one ``loss.item()`` per micro-step. ``torch.compile`` is left off: there is no Triton in this
image, so Inductor cannot generate GPU code here. GPT-2-124M, T=1024, synthetic tokens.
"""
import importlib.util
import json
import sys
import time

import torch

path = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps, warmup, T = 10, 3, 1024
spec = importlib.util.spec_from_file_location("reference_model", path)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
cfg = ref.LLMconfig(vocab_size=50304, block_size=T, n_embd=768, pos_emb="learn", up_dim=3072, non_linearity="gelu",
                    dropout=0.0, n_layer=12, moe=False, n_exp=1, n_shared=0, n_act=1, coeff=0.01, aux_free=False,
                    alpha=0.0, gamma=0.0, attn="mha", n_head=12, n_kv_heads=12, q_latent_dim=None,
                    kv_latent_dim=None, rope_head_dim=None, act_recomp=False)
torch.manual_seed(1729)
model = ref.LLM(cfg).cuda()
opt = model.configure_optimizers(weight_decay=0.1, learning_rate=3e-4, device="cuda")
ctx = torch.amp.autocast(device_type="cuda", dtype=torch.bfloat16)
x = torch.randint(0, 50304, (B, T), device="cuda")
y = torch.randint(0, 50304, (B, T), device="cuda")


def step():
    opt.zero_grad(set_to_none=True)
    with ctx:
        _, loss, _ = model(x, y)
    loss.item()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()


for _ in range(warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"reference_tokens_per_s": B * T / dt, "ms_per_step": 1000 * dt, "micro_batch": B,
                  "mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))
