"""KV-cache decoding: cached incremental logits equal full-sequence logits for every attention
type, the sliding window keeps block_size-1 positions, and sampling honours top-k."""
import pytest
import torch

from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

BASE = dict(vocab_size=97, block_size=24, n_embd=32, n_layer=2, n_head=4, up_dim=48)
VARIANTS = {
    "mha_learn": dict(attn="mha", pos_emb="learn", non_linearity="gelu"),
    "gqa_rope": dict(attn="gqa", n_kv_heads=2, pos_emb="rope", non_linearity="swiglu"),
    "mqa_sin": dict(attn="mqa", pos_emb="sin", non_linearity="gelu"),
    "mla_naive": dict(attn="mla", pos_emb="learn", q_latent_dim=8, kv_latent_dim=16),
    "mla_rope": dict(attn="mla", pos_emb="rope", q_latent_dim=8, kv_latent_dim=16, rope_head_dim=8),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_cached_decode_equals_full_forward(name):
    torch.manual_seed(0)
    cfg = LLMconfig(**BASE, **VARIANTS[name]).finalize()
    m = LLM(cfg).eval()
    idx = torch.randint(0, cfg.vocab_size, (2, 12))
    with torch.no_grad():
        logits, _, caches = m(idx[:, :8])
        steps = [logits[:, -1]]
        for t in range(8, 12):
            logits, _, caches = m(idx[:, t:t + 1], kv_caches=caches)
            steps.append(logits[:, -1])
        for i, t in enumerate(range(8, 13)):
            full, _, _ = m(idx[:, :t])
            assert torch.allclose(steps[i], full[:, -1], rtol=1e-4, atol=1e-5), f"{name}: step {t}"
    if cfg.attn != "mla":
        k, v = caches[0]
        assert k.shape == (2, cfg.n_kv_heads, 12, cfg.n_embd // cfg.n_head)   # reference cache layout


def test_generate_sliding_window_and_topk():
    torch.manual_seed(1)
    cfg = LLMconfig(**BASE, **VARIANTS["gqa_rope"]).finalize()
    m = LLM(cfg)
    out = m.generate(torch.randint(0, cfg.vocab_size, (2, 5)), max_new_tokens=40, temperature=0.8, topk=1)
    assert out.shape == (2, 45) and m.training        # generate() restores train mode
    out2 = m.generate(out[:, :5], max_new_tokens=40, temperature=0.8, topk=1)
    assert torch.equal(out, out2)                      # top-1 sampling is deterministic
