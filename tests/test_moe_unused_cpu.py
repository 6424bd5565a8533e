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


def test_row_scatter_and_combine_statements_drop_negative_rows():
    """PyTorch statement of the row scatter / combine with dropped pairs (row -1): the pair writes nothing, adds
    nothing to its token's output and passes no gradient."""
    import torch
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    N, k, C, rows = 6, 2, 4, 9
    row = torch.tensor([0, 3, -1, 4, 8, -1, 1, 2, -1, -1, 5, 6])
    x = torch.randn(N, C, requires_grad=True)
    gates = torch.rand(N, k, requires_grad=True)
    xs = ops.scatter_rows(x, row, k, rows)
    assert xs.shape == (rows, C) and (xs[7] == 0).all()
    for p in range(N * k):
        if row[p] >= 0:
            assert torch.equal(xs[row[p]], x[p // k].detach())
    out = ops.gather_combine(xs * 3.0, row, gates, k)
    keep = (row >= 0).view(N, k).float()
    want = 3.0 * (gates.detach() * keep).sum(1, keepdim=True) * x.detach()
    assert torch.allclose(out, want, atol=1e-6)
    out.sum().backward()
    assert torch.allclose(x.grad, 3.0 * (gates.detach() * keep).sum(1, keepdim=True).expand(N, C), atol=1e-6)
    assert (gates.grad[keep == 0] == 0).all()
