"""Allocated GPU memory over repeated generate() calls (graphed decode over a fresh static cache per call, varying
batch sizes and prompt lengths): growth of `allocated` between rounds is a leak.   python benchmarks/debug/generate_leak_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

for name, kw in (("gpt2-like", dict(attn="mha", pos_emb="learn", non_linearity="gelu")),
                 ("gqa-rope-swiglu", dict(attn="gqa", n_kv_heads=4, pos_emb="rope", non_linearity="swiglu")),
                 ("mla-rope", dict(attn="mla", pos_emb="rope", q_latent_dim=64, kv_latent_dim=64, rope_head_dim=32, non_linearity="swiglu"))):
    cfg = LLMconfig(vocab_size=50304, block_size=512, n_embd=768, n_layer=4, n_head=12, up_dim=2048, **kw).finalize()
    torch.manual_seed(0)
    m = LLM(cfg).to("cuda", torch.bfloat16).eval()
    for rnd in range(4):
        for B, P in ((1, 16), (8, 100), (3, 7), (16, 300)):
            idx = torch.randint(0, cfg.vocab_size, (B, P), device="cuda")
            out = m.generate(idx, max_new_tokens=64, temperature=0.8, topk=50)
            assert out.shape == (B, P + 64)
            del out, idx
        torch.cuda.synchronize()
        print(name, "round", rnd, "graphed", m.last_decode_graphed, "allocated MB %.1f" % (torch.cuda.memory_allocated() / 2**20),
              "reserved MB %.0f" % (torch.cuda.memory_reserved() / 2**20), flush=True)
    del m
