"""An expert that receives no token must still take part in the backward pass: the engines count a
gradient hook per parameter, and ranks whose hooks fire for different parameters would issue their
collectives in different orders."""
import torch

from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM


def test_experts_without_tokens_get_zero_gradients():
    torch.manual_seed(0)
    cfg = LLMconfig(vocab_size=128, block_size=16, n_embd=32, n_layer=1, n_head=2, up_dim=64, attn="mha",
                    pos_emb="learn", non_linearity="gelu", moe=True, n_exp=16, n_shared=1, n_act=2).finalize()
    model = LLM(cfg)
    idx = torch.randint(0, cfg.vocab_size, (1, 4))          # 4 tokens x 1 routed slot << 15 routed experts
    _, loss, _ = model(idx, idx)
    loss.backward()
    moe = model.transformer.h[0].moe
    counts = 0
    for e in moe.experts[moe.n_shared:]:
        g = e.expert.c_fc.weight.grad
        assert g is not None and e.expert.c_proj.weight.grad is not None
        counts += int(g.abs().sum() > 0)
    assert 0 < counts <= 4            # only the experts that saw tokens have non-zero gradients
