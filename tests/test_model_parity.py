"""Our LLM against the reference model used as an oracle (imported by path, CPU, fp32):
identical state-dict key sets and shapes, strict loading in both directions, identical logits
and loss, identical parameter counts."""
import dataclasses

import pytest
import torch

from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

BASE = dict(vocab_size=320, block_size=48, n_embd=64, n_layer=2, n_head=4, up_dim=96, dropout=0.0)
VARIANTS = {
    "gpt2": dict(attn="mha", pos_emb="learn", non_linearity="gelu"),
    "mqa_sin_relu": dict(attn="mqa", pos_emb="sin", non_linearity="relu"),
    "gqa_rope_swiglu": dict(attn="gqa", n_kv_heads=2, pos_emb="rope", non_linearity="swiglu"),
    "mla_naive": dict(attn="mla", pos_emb="learn", q_latent_dim=16, kv_latent_dim=24),
    "mla_rope": dict(attn="mla", pos_emb="rope", q_latent_dim=16, kv_latent_dim=24, rope_head_dim=8),
    "moe_aux": dict(attn="mha", pos_emb="learn", non_linearity="swiglu", moe=True, n_exp=6, n_shared=1, n_act=3),
    "moe_auxfree": dict(attn="gqa", n_kv_heads=2, pos_emb="rope", non_linearity="gelu", moe=True, n_exp=5,
                        n_shared=2, n_act=4, aux_free=True),
    "mla_rope_moe_auxfree": dict(attn="mla", pos_emb="rope", q_latent_dim=16, kv_latent_dim=24, rope_head_dim=8,
                                 non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2, aux_free=True),
    "gpt2_mish": dict(attn="mha", pos_emb="learn", non_linearity="mish"),
    "gqa_sin_lrelu": dict(attn="gqa", n_kv_heads=1, pos_emb="sin", non_linearity="lrelu"),
    "mha_rope_tanh": dict(attn="mha", pos_emb="rope", non_linearity="tanh"),
    "mla_naive_unknown_act": dict(attn="mla", pos_emb="sin", q_latent_dim=16, kv_latent_dim=24,
                                  non_linearity="no_such_activation"),      # falls back to GELU like the reference
}


def _ref_config(ref_mod, cfg: LLMconfig):
    names = [f.name for f in dataclasses.fields(ref_mod.LLMconfig)]
    return ref_mod.LLMconfig(**{n: getattr(cfg, n) for n in names})


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_state_dict_and_numerics_match_reference(name, reference_model_module):
    ref_mod = reference_model_module
    torch.manual_seed(0)
    cfg = LLMconfig(**BASE, **VARIANTS[name]).finalize()
    ours = LLM(cfg)
    ours.materialize_logits = True
    ref = ref_mod.LLM(_ref_config(ref_mod, cfg))
    sd_ours, sd_ref = ours.state_dict(), ref.state_dict()
    assert list(sd_ours.keys()) == list(sd_ref.keys())
    for k in sd_ref:
        assert sd_ours[k].shape == sd_ref[k].shape and sd_ours[k].dtype == sd_ref[k].dtype, k
    ref.load_state_dict(sd_ours, strict=True)         # ours -> reference
    assert ours.get_num_params() == ref.get_num_params()
    idx = torch.randint(0, cfg.vocab_size, (3, 40))
    tgt = torch.randint(0, cfg.vocab_size, (3, 40))
    tgt[0, :5] = -1
    ours.train(); ref.train()
    if cfg.moe and cfg.aux_free:
        bias0 = [b.clone() for n, b in ours.named_buffers() if n.endswith("expert_bias")]
    lo, loss_o, _ = ours(idx, tgt)
    lr_, loss_r, _ = ref(idx, tgt)
    assert torch.allclose(lo, lr_, rtol=1e-4, atol=1e-5)
    assert abs(loss_o.item() - loss_r.item()) < 1e-5
    loss_o.backward(); loss_r.backward()
    go = dict(ours.named_parameters())
    for n, p in ref.named_parameters():
        assert torch.allclose(go[n].grad, p.grad, rtol=1e-3, atol=1e-6), f"grad {n}"
    if cfg.moe and cfg.aux_free:   # the aux-free bias moved identically
        for (n, b), (_, br) in zip([(n, b) for n, b in ours.named_buffers() if n.endswith("expert_bias")],
                                   [(n, b) for n, b in ref.named_buffers() if n.endswith("expert_bias")]):
            assert torch.allclose(b, br, atol=1e-7)
    # reference -> ours
    torch.manual_seed(1)
    ref2 = ref_mod.LLM(_ref_config(ref_mod, cfg))
    ours.load_state_dict(ref2.state_dict(), strict=True)
    ours.eval(); ref2.eval()
    ours.VAL_RUN = ref2.VAL_RUN = True
    with torch.no_grad():
        _, l1, _ = ours(idx, tgt)
        _, l2, _ = ref2(idx, tgt)
    assert abs(l1.item() - l2.item()) < 1e-5


@pytest.mark.parametrize("preset,count", [("gpt2-124m", 124_429_824), ("gpt2-350m", 354_748_416)])
def test_exact_parameter_counts(preset, count):
    from distributed_pytorch_amd.config import PRESETS
    cfg = LLMconfig(**{**LLMconfig().to_dict(), **PRESETS[preset]}).finalize()
    with torch.device("meta"):
        m = LLM(cfg)
    assert sum(p.numel() for p in m.parameters()) == count


def test_fused_loss_equals_materialized_loss():
    torch.manual_seed(3)
    cfg = LLMconfig(**BASE, **VARIANTS["gpt2"]).finalize()
    m = LLM(cfg)
    idx = torch.randint(0, cfg.vocab_size, (2, 33))
    _, a, _ = m(idx, idx)
    m.materialize_logits = True
    lg, b, _ = m(idx, idx)
    assert lg.shape == (2, 33, cfg.vocab_size) and abs(a.item() - b.item()) < 1e-6
    lg2, none_loss, caches = m(idx)          # inference: last-token logits only
    assert lg2.shape == (2, 1, cfg.vocab_size) and none_loss is None and len(caches) == cfg.n_layer
