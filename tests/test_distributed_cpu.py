"""DDP / ZeRO-1 / ZeRO-2 / FSDP on 2 gloo ranks must reproduce the single-process trajectory on
the same global batches (deterministic per-rank data sharding makes that an equality test)."""
import copy

import pytest
import torch

from dist_utils import run_distributed

from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

CFGS = {
    "gpt": dict(vocab_size=256, block_size=32, n_embd=64, n_layer=3, n_head=4, up_dim=128, attn="mha",
                pos_emb="learn", non_linearity="gelu"),
    "gqa_rope_swiglu": dict(vocab_size=256, block_size=32, n_embd=64, n_layer=2, n_head=4, n_kv_heads=2, up_dim=96,
                            attn="gqa", pos_emb="rope", non_linearity="swiglu"),
}


def _make(cfg_name, seed=0):
    torch.manual_seed(seed)
    cfg = LLMconfig(**CFGS[cfg_name]).finalize()
    return LLM(cfg), cfg


def _batches(cfg, n_steps, global_b, accum):
    g = torch.Generator().manual_seed(123)
    return [[torch.randint(0, cfg.vocab_size, (global_b, cfg.block_size), generator=g) for _ in range(accum)]
            for _ in range(n_steps)]


def _train(engine, batches, rank, world, accum):
    losses = []
    for step_batches in batches:
        engine.zero_grad()
        for mi, xb in enumerate(step_batches):
            per = xb.shape[0] // world
            x = xb[rank * per:(rank + 1) * per]
            engine.require_backward_grad_sync = (mi == accum - 1)
            _, loss, _ = engine(x, x.roll(-1, dims=1))
            (loss / accum).backward()
            losses.append(loss.item())
        engine.clip_grad_norm_(1.0)
        engine.step(1e-2)
    return losses


def _single_reference(cfg_name, n_steps=3, global_b=4, accum=2):
    from distributed_pytorch_amd.parallel.engine import Engine
    model, cfg = _make(cfg_name)
    eng = Engine(model, weight_decay=0.1, dtype="fp32", bucket_mb=0.05, eps=1e-5)
    _train(eng, _batches(cfg, n_steps, global_b, accum), 0, 1, accum)
    return {k: v.clone() for k, v in eng.state_dict().items()}


def _worker(rank, world, mode, cfg_name, n_steps, global_b, accum):
    model, cfg = _make(cfg_name, seed=rank)      # different init per rank: the engine must sync from rank 0
    if rank == 0:
        model, cfg = _make(cfg_name, seed=0)
    kw = dict(weight_decay=0.1, dtype="fp32", bucket_mb=0.05, eps=1e-5)
    if mode.endswith("_vp"):      # head + loss with the vocabulary split over the ranks
        kw["vocab_parallel"] = True
        mode = mode[:-3]
    if mode == "ddp":
        from distributed_pytorch_amd.parallel.ddp import DDPEngine
        eng = DDPEngine(model, **kw)
    elif mode in ("zero1", "zero2"):
        from distributed_pytorch_amd.parallel.zero import ZeroEngine
        eng = ZeroEngine(model, stage=int(mode[-1]), **kw)
    elif mode == "fsdp":
        from distributed_pytorch_amd.parallel.fsdp import FSDPEngine
        eng = FSDPEngine(model, **kw)
    elif mode == "fsdp_keep":
        from distributed_pytorch_amd.parallel.fsdp import FSDPEngine
        eng = FSDPEngine(model, reshard_after_forward=False, **kw)
    elif mode.startswith("fsdp_p"):           # fsdp_p<prefetch>i<inflight>: the tunables of --fsdp_prefetch / --fsdp_inflight
        from distributed_pytorch_amd.parallel.fsdp import FSDPEngine
        pf, fl = mode[len("fsdp_p"):].split("i")
        eng = FSDPEngine(model, prefetch=int(pf), inflight=int(fl), **kw)
        assert eng.prefetch == int(pf) and eng.inflight == int(fl)
    assert (eng.module.vocab_parallel is not None) == ("vocab_parallel" in kw)
    _train(eng, _batches(cfg, n_steps, global_b, accum), rank, world, accum)
    sd = eng.state_dict()
    return {k: v.clone() for k, v in sd.items()} if rank == 0 else None


@pytest.mark.parametrize("mode", ["ddp", "zero1", "zero2", "fsdp", "fsdp_keep", "ddp_vp", "zero2_vp", "fsdp_vp"])
@pytest.mark.parametrize("cfg_name", ["gpt", "gqa_rope_swiglu"])
def test_matches_single_process(mode, cfg_name):
    if cfg_name != "gpt" and mode in ("zero1", "fsdp_keep", "zero2_vp", "fsdp_vp"):
        pytest.skip("covered by the gpt config")
    want = _single_reference(cfg_name)
    got = run_distributed(_worker, 2, mode, cfg_name, 3, 4, 2)[0]
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if want[k].is_complex():
            assert torch.allclose(torch.view_as_real(got[k]), torch.view_as_real(want[k]))
            continue
        assert torch.allclose(got[k].float(), want[k].float(), rtol=2e-4, atol=2e-5), f"{mode}: {k} differs"


@pytest.mark.parametrize("mode", ["fsdp_p0i1", "fsdp_p2i3", "fsdp_p3i1"])
def test_fsdp_prefetch_and_inflight_settings_match_single_process(mode):
    """No gather ahead and one reduce-scatter in flight, two / three units gathered ahead with up to three
    reduce-scatters in flight: the same trajectory as one process in every setting."""
    want = _single_reference("gpt")
    got = run_distributed(_worker, 2, mode, "gpt", 3, 4, 2)[0]
    for k in want:
        assert torch.allclose(got[k].float(), want[k].float(), rtol=2e-4, atol=2e-5), f"{mode}: {k} differs"


@pytest.mark.parametrize("mode", ["ddp", "zero2", "fsdp", "ddp_vp", "fsdp_vp"])
def test_four_ranks_match_single_process(mode):
    """World size 4 (shards of a quarter, uneven bucket tails, two prefetches in flight): same
    trajectory as one process on the same global batches."""
    want = _single_reference("gpt", n_steps=2, global_b=8, accum=1)
    got = run_distributed(_worker, 4, mode, "gpt", 2, 8, 1)[0]
    for k in want:
        assert torch.allclose(got[k].float(), want[k].float(), rtol=2e-4, atol=2e-5), f"{mode}: {k} differs"


@pytest.mark.parametrize("mode", ["ddp", "fsdp", "ddp_vp", "fsdp_vp"])
def test_eight_ranks_match_single_process(mode):
    """The world size of one MI355X node."""
    want = _single_reference("gpt", n_steps=2, global_b=8, accum=1)
    got = run_distributed(_worker, 8, mode, "gpt", 2, 8, 1)[0]
    for k in want:
        assert torch.allclose(got[k].float(), want[k].float(), rtol=2e-4, atol=2e-5), f"{mode}/8: {k} differs"


def _collective_bytes_worker(rank, world):
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.parallel.zero import ZeroEngine
    model, cfg = _make("gpt")
    for k in comm.STATS:
        comm.STATS[k] = 0
    eng = ZeroEngine(model, stage=2, weight_decay=0.1, dtype="fp32", bucket_mb=0.05, eps=1e-5)
    init_bcast = comm.STATS["broadcast"]
    x = torch.randint(0, cfg.vocab_size, (2, cfg.block_size))
    eng.zero_grad()
    _, loss, _ = eng(x, x)
    loss.backward()
    eng.step(1e-3)
    total = sum(g.numel for g in eng.groups) * 4
    return dict(stats=dict(comm.STATS), total=total, init_bcast=init_bcast, ngroups=len(eng.groups))


def test_zero2_uses_reduce_scatter_and_all_gather_only():
    out = run_distributed(_collective_bytes_worker, 2)[0]
    st = out["stats"]
    assert out["ngroups"] >= 3
    assert st["reduce_scatter"] == out["total"], "every gradient byte goes through reduce-scatter exactly once"
    assert st["all_gather"] == out["total"], "every parameter byte comes back through all-gather exactly once"
    assert st["all_reduce"] <= 64, "only the scalar norm is all-reduced"


def _vp_worker(rank, world):
    import torch.nn.functional as F
    from distributed_pytorch_amd.ops import vocab_parallel_cross_entropy
    torch.manual_seed(0)
    N, V = 24, 40
    logits = torch.randn(N, V) * 3
    targets = torch.randint(0, V, (N,))
    targets[::5] = -1                                    # ignored rows
    ref_in = logits.clone().requires_grad_()
    ref = F.cross_entropy(ref_in, targets, ignore_index=-1)
    ref.backward()
    per = V // world
    local = logits[:, rank * per:(rank + 1) * per].clone().requires_grad_()
    loss = vocab_parallel_cross_entropy(local, targets, rank * per)
    loss.backward()
    return loss.item(), ref.item(), (local.grad - ref_in.grad[:, rank * per:(rank + 1) * per]).abs().max().item()


def test_vocab_parallel_cross_entropy_matches_full():
    """Logits sharded along the vocabulary over 2 ranks: same loss and gradient as the full
    cross-entropy (max / sum-exp / target logit combined with two small all-reduces)."""
    for loss, ref, gerr in run_distributed(_vp_worker, 2):
        assert abs(loss - ref) < 1e-5
        assert gerr < 1e-6


def _vpl_worker(rank, world):
    import torch.nn.functional as F
    from distributed_pytorch_amd.ops import vocab_parallel_linear_cross_entropy
    from distributed_pytorch_amd.ops.loss import vocab_slices
    torch.manual_seed(0)
    N, C, V = 10, 16, 40
    xs = torch.randn(world, N, C)
    w = torch.randn(V, C) * 0.3
    ts = torch.randint(0, V, (world, N))
    ts[:, ::4] = -1
    # reference: the global-batch loss on one process; a rank's gradients are W times its share (the engines average)
    xr = xs.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    ref = F.cross_entropy((xr.view(-1, C) @ wr.t()), ts.view(-1), ignore_index=-1)
    ref.backward()
    x = xs[rank].clone().requires_grad_()
    wl = w.clone().requires_grad_()
    loss = vocab_parallel_linear_cross_entropy(x, wl, ts[rank], chunk_rows=8)     # 30 gathered rows: 4 chunks, 2 buffers
    loss.backward()
    lo, hi = vocab_slices(V, world)[rank]
    own = torch.zeros_like(w)
    own[lo:hi] = wr.grad[lo:hi] * world
    return (abs(loss.item() - ref.item()), (x.grad - xr.grad[rank] * world).abs().max().item(),
            (wl.grad - own).abs().max().item(), (lo, hi))


def test_vocab_parallel_head_with_uneven_slices_and_pipelined_chunks():
    """Three ranks, a vocabulary of 40 (slices 14 / 14 / 12), row chunks smaller than the gathered batch: the loss is
    the global one, dx the rank's rows of the global gradient, dw this rank's vocabulary rows of it (times W)."""
    from distributed_pytorch_amd.ops.loss import vocab_slices
    assert vocab_slices(40, 3) == [(0, 14), (14, 28), (28, 40)]
    assert vocab_slices(50304, 8) == [(i * 6400, min(50304, (i + 1) * 6400)) for i in range(8)]
    assert vocab_slices(50304, 1) == [(0, 50304)]
    for dl, dxe, dwe, _ in run_distributed(_vpl_worker, 3):
        assert dl < 1e-5 and dxe < 1e-5 and dwe < 1e-5


def _launch_order_worker(rank, world):
    """Record the order in which the DDP buckets are reduced and how many had left when backward ended."""
    from distributed_pytorch_amd.parallel.ddp import DDPEngine
    model, cfg = _make("gpt")
    eng = DDPEngine(model, weight_decay=0.1, dtype="fp32", bucket_mb=0.05, eps=1e-5)
    order = []
    orig = eng._reduce_group
    eng._reduce_group = lambda gi: (order.append(gi), orig(gi))[1]
    x = torch.randint(0, cfg.vocab_size, (2, cfg.block_size))
    eng.zero_grad()
    _, loss, _ = eng(x, x)
    loss.backward()
    in_backward = len(order)
    eng.step(1e-3)
    return dict(order=order, in_backward=in_backward, ngroups=len(eng.groups))


def test_ddp_buckets_leave_last_to_first_inside_backward():
    """Overlap and determinism: every bucket is reduced from a gradient hook (during backward), in
    strictly descending order on every rank."""
    for out in run_distributed(_launch_order_worker, 2):
        n = out["ngroups"]
        assert n >= 3 and out["order"] == list(range(n - 1, -1, -1)), out
        assert out["in_backward"] == n, out


# ---- the tied lm_head / tkn_emb gradient: head contribution reduced early ---------------------------------
def _split_tail_worker(rank, world, accum):
    from distributed_pytorch_amd.parallel.ddp import DDPEngine
    model, cfg = _make("gpt", seed=rank if rank else 0)
    eng = DDPEngine(model, weight_decay=0.1, dtype="fp32", bucket_mb=0.05, eps=1e-5, tied_tail="split")
    assert eng._head_sink is not None and eng.module.head_weight_hook is not None
    logs = []
    batches = _batches(cfg, 3, 4, accum)
    for step_batches in batches:
        eng.zero_grad()
        for mi, xb in enumerate(step_batches):
            per = xb.shape[0] // world
            x = xb[rank * per:(rank + 1) * per]
            eng.require_backward_grad_sync = (mi == accum - 1)
            _, loss, _ = eng(x, x.roll(-1, dims=1))
            (loss / accum).backward()
            if mi == accum - 1:
                logs.append((list(eng.launch_log), eng._head_work is not None))
        eng.clip_grad_norm_(1.0)
        eng.step(1e-2)
    sd = eng.state_dict()
    return ({k: v.clone() for k, v in sd.items()} if rank == 0 else None), logs, len(eng.groups)


@pytest.mark.parametrize("accum", [1, 2])
def test_split_tied_tail_matches_and_head_leaves_first(accum):
    """tied_tail="split": same trajectory as one process, and on the synchronising micro-step the head's
    share of the tied gradient is the FIRST collective of backward (before the last block's bucket)
    while the bucket holding the embedding is the last."""
    want = _single_reference("gpt", n_steps=3, global_b=4, accum=accum)
    outs = run_distributed(_split_tail_worker, 2, accum)
    got = outs[0][0]
    for k in want:
        assert torch.allclose(got[k].float(), want[k].float(), rtol=2e-4, atol=2e-5), f"split tail: {k} differs"
    for _, logs, ngroups in outs:
        for log, head_in_flight in logs:
            assert head_in_flight
            assert log[0] == "head" and log[1] == ngroups - 1 and log[-1] == 0, log
            assert log[1:] == list(range(ngroups - 1, -1, -1)), log


# ---- fp32 gradient buffers on bf16 parameters (--grad_dtype fp32) -------------------------------------
def _wide_worker(rank, world, mode, grad_dtype, accum):
    model, cfg = _make("gpt", seed=0)
    kw = dict(weight_decay=0.1, dtype="bf16", grad_dtype=grad_dtype, bucket_mb=0.05, eps=1e-5)
    if mode == "single":
        from distributed_pytorch_amd.parallel.engine import Engine
        eng = Engine(model, **kw)
    elif mode == "ddp":
        from distributed_pytorch_amd.parallel.ddp import DDPEngine
        eng = DDPEngine(model, **kw)
    elif mode == "zero2":
        from distributed_pytorch_amd.parallel.zero import ZeroEngine
        eng = ZeroEngine(model, stage=2, **kw)
    else:
        from distributed_pytorch_amd.parallel.fsdp import FSDPEngine
        eng = FSDPEngine(model, **kw)
    g0 = eng.groups[-1]
    init = {s_.name: eng.optims[gi].master.clone() for gi, g in enumerate(eng.groups) for s_ in g.slots} \
        if mode == "single" else None
    assert g0.wide_grad == (grad_dtype == "fp32")
    if grad_dtype == "fp32" and mode != "fsdp":      # (sharded units materialise their buffers in backward)
        assert g0.rgrad.dtype == torch.float32 and g0.grad.dtype == torch.bfloat16
    _train(eng, _batches(cfg, 2, 4, accum), rank, world, accum)
    # masters are the fp32 truth of the trajectory
    out = {}
    for gi, g in enumerate(eng.groups):
        full = eng._full_master(gi) if hasattr(eng, "_full_master") else eng.optims[gi].master
        for s in g.slots:
            out[s.name] = full[s.offset:s.offset + s.numel].clone()
    if mode == "single":
        out["__init__"] = torch.cat([init[k].reshape(-1) for k in sorted(init)])
    return out if rank == 0 else None


@pytest.mark.parametrize("mode", ["ddp", "zero2", "fsdp"])
def test_fp32_gradient_buffers_match_single_process(mode):
    """bf16 parameters with fp32 gradient accumulation and reduction: every engine on two ranks follows
    the single-process engine with the same setting (same global batches, two micro-steps)."""
    want = _wide_worker(0, 1, "single", "fp32", 2)
    init = want.pop("__init__")
    got = run_distributed(_wide_worker, 2, mode, "fp32", 2)[0]
    # AdamW turns a rounding-level difference of a near-zero gradient into a full-size step of that
    # element, so the trajectories are compared as a whole: the distance between the two runs against
    # the distance either has travelled from the initial weights.
    w = torch.cat([want[k].reshape(-1) for k in sorted(want)])
    g = torch.cat([got[k].reshape(-1) for k in sorted(want)])
    travelled = (w - init[: w.numel()]).norm().item() if init.numel() == w.numel() else w.norm().item()
    assert (g - w).norm().item() < 0.15 * travelled, (mode, (g - w).norm().item(), travelled)


def test_fp32_gradient_accumulation_drifts_less_than_bf16():
    """8 micro-steps accumulated into one optimizer step: the update from fp32 gradient buffers is closer
    to an all-fp32 run than the update from bf16 buffers, whose relative error stays bounded."""
    from distributed_pytorch_amd.parallel.engine import Engine
    accum = 8

    def grads(dtype, grad_dtype):
        model, cfg = _make("gpt", seed=0)
        eng = Engine(model, weight_decay=0.0, dtype=dtype, grad_dtype=grad_dtype, bucket_mb=0.05)
        eng.zero_grad()
        for xb in _batches(cfg, 1, 4, accum)[0]:
            _, loss, _ = eng(xb, xb.roll(-1, dims=1))
            (loss / accum).backward()
        eng._fold_grads()
        return torch.cat([g.rgrad.float() for g in eng.groups])

    ref = grads("fp32", "auto")
    wide = grads("bf16", "fp32")
    narrow = grads("bf16", "bf16")
    err_wide = ((wide - ref).norm() / ref.norm()).item()
    err_narrow = ((narrow - ref).norm() / ref.norm()).item()
    # both carry the bf16 forward / backward error; only the narrow one adds accumulation rounding
    assert err_wide <= err_narrow * 1.02, (err_wide, err_narrow)
    assert err_narrow < 5e-2, err_narrow


# ---- FSDP buffer lifetime ----------------------------------------------------------------------------
def _fsdp_lifetime_worker(rank, world):
    import os
    from distributed_pytorch_amd.parallel.fsdp import FSDPEngine
    os.environ["DPA_DEBUG_STREAMS"] = "1"      # the whole test runs with the ordering assertions armed
    model, cfg = _make("gpt", seed=0)
    eng = FSDPEngine(model, weight_decay=0.1, dtype="fp32", bucket_mb=0.05, eps=1e-5)
    u = eng.block_units[1]
    assert not u.gathered
    # a prefetch that nobody consumes (e.g. an evaluation that stops early): releasing the unit retires the
    # gather first and leaves it in a state from which the next forward gathers again
    u.begin_gather(eng.group)
    assert u.gather_work is not None
    u.release_params()
    assert u.gather_work is None and not u.gathered and u.data.untyped_storage().nbytes() == 0
    batches = _batches(cfg, 2, 4, 1)
    losses = _train(eng, batches, rank, world, 1)      # full steps under the assertions: none fires
    assert all(torch.isfinite(torch.tensor(l)) for l in losses)
    # releasing a gradient buffer under a reduce-scatter that has not been waited for is caught
    class _Live:
        def wait(self):
            return True
    u.materialize_grad()
    u.rs_work = _Live()
    try:
        u.release_grad()
        raised = False
    except AssertionError:
        raised = True
    u.rs_work = None
    u.release_grad()
    os.environ.pop("DPA_DEBUG_STREAMS")
    return raised


def test_fsdp_unconsumed_prefetch_and_release_ordering():
    assert all(run_distributed(_fsdp_lifetime_worker, 2))
