"""Which pieces of a decode step can be captured in a hipGraph (debugging aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from distributed_pytorch_amd import ops
from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM, StaticKVCache

dev = "cuda"


def try_capture(name, fn):
    try:
        fn()
        torch.cuda.synchronize()
    except Exception as e:
        print("EAGER-FAIL", name, type(e).__name__, str(e).splitlines()[0][:120], flush=True)
        return
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        print("OK  ", name, flush=True)
    except Exception as e:
        torch.cuda.synchronize()
        print("FAIL", name, type(e).__name__, str(e).splitlines()[0][:120], flush=True)


torch.manual_seed(0)
cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                pos_emb="learn", non_linearity="gelu").finalize()
model = LLM(cfg).to(dev).to(torch.bfloat16).eval()
B = 3
with torch.no_grad():
    idx = torch.randint(0, 1024, (B, 17), device=dev)
    logits, _, caches = model(idx)
    static = StaticKVCache(cfg, B, cfg.block_size, dev, torch.bfloat16)
    static.fill(caches, 17)
    tok = torch.randint(0, 1024, (B, 1), device=dev)
    wte = model.lm_head.weight
    x = torch.randn(B, 1, 256, device=dev, dtype=torch.bfloat16)
    blk = model.transformer.h[0]
    try_capture("embedding+pos", lambda: ops.embedding(tok, wte, model.pos_emb.weight.index_select(0, static.pos), 0))
    try_capture("layer_norm", lambda: ops.layer_norm(x, blk.ln1.weight, blk.ln1.bias, 1e-5))
    try_capture("add_layer_norm", lambda: ops.add_layer_norm(x, x, blk.ln1.weight, blk.ln1.bias, 1e-5))
    try_capture("linear qkv (M=3)", lambda: ops.linear(x, blk.attn.attn.c_attn.weight, blk.attn.attn.c_attn.bias))
    q = torch.randn(B, 1, 4, 64, device=dev, dtype=torch.bfloat16)
    try_capture("cached_attention", lambda: ops.cached_attention(q, static.k[0], static.v[0], static.len_after))
    h = torch.randn(B, 1, 512, device=dev, dtype=torch.bfloat16)
    try_capture("gelu", lambda: ops.gelu(h))
    try_capture("attn module", lambda: blk.attn(x, None, static.layer(0), False))
    try_capture("mlp module", lambda: blk.mlp(x))
    try_capture("block", lambda: blk(x, None, None, static.layer(0), False))
    try_capture("lm_head", lambda: ops.linear(x[:, -1:, :], wte))
    try_capture("model.forward", lambda: model(tok, kv_caches=static))
    try_capture("sample", lambda: LLM._sample(logits, 1.0, 1))
