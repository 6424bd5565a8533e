"""The residual adds carried by the branch GEMMs (``DPA_FUSE_RESIDUAL``) give the same model as the
adds carried by the LayerNorm kernels: loss and every gradient, with and without recomputation."""
import pytest
import torch

from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM


def _run(monkeypatch, mode, **over):
    monkeypatch.setenv("DPA_FUSE_RESIDUAL", mode)
    torch.manual_seed(0)
    kw = dict(vocab_size=96, block_size=16, n_embd=32, n_layer=3, n_head=4, n_kv_heads=2, up_dim=64, attn="gqa",
              pos_emb="rope", non_linearity="swiglu", dropout=0.0)
    kw.update(over)
    cfg = LLMconfig(**{k: v for k, v in kw.items() if k in LLMconfig.__dataclass_fields__})
    if hasattr(cfg, "finalize"):
        cfg.finalize()
    model = LLM(cfg)
    model.train()
    idx = torch.randint(0, cfg.vocab_size, (2, 16))
    tgt = torch.randint(0, cfg.vocab_size, (2, 16))
    out = model(idx, tgt)
    loss = out[1]
    loss.backward()
    return loss.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("over", [dict(), dict(non_linearity="gelu", pos_emb="learn", attn="mha"),
                                  dict(act_recomp=True), dict(attn="mla", q_latent_dim=16, kv_latent_dim=16, rope_head_dim=4),
                                  dict(moe=True, n_exp=4, n_shared=1, n_act=2)])
def test_fused_residual_equals_layernorm_adds(monkeypatch, over):
    try:
        l0, g0 = _run(monkeypatch, "0", **over)
    except TypeError as e:   # a config field of another name in this tree
        pytest.skip(str(e))
    l1, g1 = _run(monkeypatch, "force", **over)
    assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-6)
    assert g0.keys() == g1.keys()
    for n in g0:
        assert torch.allclose(g0[n], g1[n], rtol=1e-4, atol=1e-6), n
