"""A restart on another number of ranks: the resume files of a W-rank job are read by a W'-rank job, every rank cutting its
own optimizer slice out of the old shards. The restored job holds exactly the weights and moments the old job had (its
gathered state dict is equal), and trains on from the same optimizer step."""
import os

import pytest
import torch

from dist_utils import run_distributed

from distributed_pytorch_amd.checkpoint import find_resume, load_resume, save_resume
from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

CFG = dict(vocab_size=128, block_size=16, n_embd=32, n_layer=2, n_head=2, up_dim=64, attn="gqa", n_kv_heads=1, pos_emb="rope",
           non_linearity="swiglu")


def _engine(mode):
    torch.manual_seed(0)
    model = LLM(LLMconfig(**CFG).finalize())
    kw = dict(weight_decay=0.1, dtype="fp32", bucket_mb=0.01)
    if mode == "zero2":
        from distributed_pytorch_amd.parallel.zero import ZeroEngine
        return ZeroEngine(model, stage=2, **kw)
    if mode == "fsdp":
        from distributed_pytorch_amd.parallel.fsdp import FSDPEngine
        return FSDPEngine(model, **kw)
    from distributed_pytorch_amd.parallel.ddp import DDPEngine
    return DDPEngine(model, **kw)


def _step(eng, rank, world, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, CFG["vocab_size"], (4, CFG["block_size"]), generator=g)
    per = 4 // world
    x = x[rank * per:(rank + 1) * per]
    eng.zero_grad()
    _, loss, _ = eng(x, x.roll(-1, dims=1))
    loss.backward()
    eng.clip_grad_norm_(1.0)
    eng.step(1e-2)
    return loss.item()


def _moments(eng):
    """Second moments of every optimizer range of this rank, as (group, lo, tensor)."""
    return [(gi, o.lo, o.exp_avg_sq.clone()) for gi, o in enumerate(eng.optims)]


def _writer(rank, world, root, mode):
    path = os.path.join(root, "run_resume")
    eng = _engine(mode)
    for it in range(3):
        _step(eng, rank, world, it)
    save_resume(path, eng, 2, (), rank, world)
    sd = eng.state_dict()                        # collective: gathered fp32 weights
    if rank == 0:
        torch.save({"sd": sd, "step": eng.step_count}, os.path.join(root, "truth.pt"))
    torch.save(_moments(eng), os.path.join(root, f"moments.{rank}.pt"))
    loss = _step(eng, rank, world, 3)            # the step the restarted job will repeat
    return loss


def _reader(rank, world, root, mode, wrote):
    path = os.path.join(root, "run_resume")
    eng = _engine(mode)
    assert find_resume(path, rank, eng.device) == 2
    assert load_resume(path, eng, (), rank) == 2
    truth = torch.load(os.path.join(root, "truth.pt"), weights_only=False)
    assert eng.step_count == truth["step"]
    sd = eng.state_dict()
    for k, v in truth["sd"].items():
        assert torch.equal(sd[k], v), k
    # every element of this rank's moment ranges equals the element of the old shard that held it
    old = [m for r in range(wrote) for m in torch.load(os.path.join(root, f"moments.{r}.pt"), weights_only=False)]
    for gi, lo, mine in _moments(eng):
        seen = torch.zeros_like(mine, dtype=torch.bool)
        for ogi, olo, om in old:
            if ogi != gi:
                continue
            a, b = max(lo, olo), min(lo + mine.numel(), olo + om.numel())
            if b > a:
                assert torch.equal(mine[a - lo:b - lo], om[a - olo:b - olo])
                seen[a - lo:b - lo] = True
        assert (mine[~seen] == 0).all()          # what no old shard covered is padding
    loss = _step(eng, rank, world, 3)
    save_resume(path, eng, 3, (), rank, world)   # the next generation is written (and committed) by the new world
    return loss


@pytest.mark.parametrize("mode,wrote,reads", [("zero2", 1, 2), ("zero2", 2, 4), ("fsdp", 2, 1), ("ddp", 2, 1)])
def test_resume_on_another_world_size(tmp_path, mode, wrote, reads):
    root = str(tmp_path)
    before = run_distributed(_writer, wrote, root, mode)
    after = run_distributed(_reader, reads, root, mode, wrote)
    # the repeated step sees the same global batch and the same state: the mean loss over the ranks agrees
    assert abs(sum(before) / len(before) - sum(after) / len(after)) < 1e-4
    import json
    m = json.load(open(os.path.join(root, "run_resume.latest")))
    assert m["iter"] == 3 and m["world"] == reads and m["prev"] == 2 and m["prev_world"] == wrote


MOE = {**CFG, "moe": True, "n_exp": 5, "n_shared": 1, "n_act": 2}     # four routed experts: 4, 2 or 1 per rank


def _ep_engine(kind):
    from distributed_pytorch_amd.parallel.ep import EPEngine, EPFSDPEngine
    torch.manual_seed(0)
    cls = EPEngine if kind == "ep" else EPFSDPEngine
    return cls(LLM(LLMconfig(**MOE).finalize()), weight_decay=0.1, dtype="fp32", bucket_mb=0.01)


def _ep_writer(rank, world, root, kind):
    path = os.path.join(root, "run_resume")
    eng = _ep_engine(kind)
    for it in range(3):
        _step(eng, rank, world, it)
    save_resume(path, eng, 2, (), rank, world)
    sd = eng.state_dict()
    if rank == 0:
        torch.save({"sd": sd, "step": eng.step_count}, os.path.join(root, "truth.pt"))
    return _step(eng, rank, world, 3)


def _ep_reader(rank, world, root, kind):
    path = os.path.join(root, "run_resume")
    eng = _ep_engine(kind)
    assert load_resume(path, eng, (), rank) == 2
    truth = torch.load(os.path.join(root, "truth.pt"), weights_only=False)
    assert eng.step_count == truth["step"]
    sd = eng.state_dict()
    for k, v in truth["sd"].items():
        assert torch.equal(sd[k].cpu(), v.cpu()), k
    return _step(eng, rank, world, 3)


@pytest.mark.parametrize("kind,wrote,reads", [("ep", 2, 1), ("ep", 2, 4), ("ep_fsdp", 1, 2)])
def test_expert_parallel_resume_on_another_world_size(tmp_path, kind, wrote, reads):
    """The routed experts change owners with the world size: their optimizer state is saved with the parameter names
    and every new rank collects its experts' weights and moments from whichever old rank held them; the dense part is
    re-cut by position like in the other engines. Equal gathered weights, and the repeated step gives the same loss
    (the experts' second moments enter it through the update that precedes it)."""
    root = str(tmp_path)
    before = run_distributed(_ep_writer, wrote, root, kind)
    after = run_distributed(_ep_reader, reads, root, kind)
    # (the load-balancing term is a function of each rank's own tokens, so the mean over ranks moves by ~1e-4 with the split)
    assert abs(sum(before) / len(before) - sum(after) / len(after)) < 1e-3
