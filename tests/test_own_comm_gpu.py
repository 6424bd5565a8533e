"""The framework's own RCCL communicator (csrc/comm/rccl_comm.cpp) on one GPU.

RCCL refuses two ranks on one device, so what a one-GPU box can check is everything but the wire: the
communicator is created from a unique id, every collective runs through RCCL's kernels on the own
stream, the stream-ordering contract holds (inputs produced on the compute stream are seen, results
are visible after work.wait() without a host block), argument checks fire on the host, and a full
training step gives the same numbers as without any communicator.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def own():
    from distributed_pytorch_amd import _ext
    assert _ext.available()
    cls = torch.classes.dpa.RcclComm
    uid = cls.unique_id()
    assert uid.dtype == torch.uint8 and uid.numel() == 128 and int(uid.ne(0).sum()) > 0
    c = cls(uid, 0, 1, 0, 0, 0, 8)
    yield c
    c.destroy()


def test_communicator_identity(own):
    assert own.rank() == 0 and own.world() == 1 and own.comm_count() == 1
    assert own.rccl_version() >= 20000
    assert own.async_error() == ""
    assert own.priority() == 0


def test_collectives_follow_the_compute_stream(own):
    """The input is produced by a long chain of kernels on the compute stream right before the call; the
    result is consumed right after work.wait(): both orderings are device-side."""
    torch.manual_seed(0)
    n = 1 << 22
    x = torch.randn(n, device="cuda")
    for dtype in (torch.float32, torch.bfloat16):
        t = x.to(dtype)
        for _ in range(20):
            t = t * 1.0001 + 0.5
        want = t.clone()
        w = own.all_reduce(t, 0)
        assert w.wait() is True
        got = t + 0          # consumer on the compute stream
        torch.cuda.synchronize()
        assert w.is_completed()
        assert torch.equal(got, want)
    # avg / max on one rank are the identity as well
    t = torch.arange(1000, device="cuda", dtype=torch.float32)
    own.all_reduce(t, 4).wait()
    own.all_reduce(t, 2).wait()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))


def test_gather_scatter_broadcast_alltoall(own):
    a = torch.randn(4096, device="cuda").bfloat16()
    out = torch.empty_like(a)
    own.all_gather(out, a).wait()
    assert torch.equal(out, a)
    own.all_gather(a, a).wait()                 # in place
    rs = torch.empty_like(a)
    own.reduce_scatter(rs, a, 0).wait()
    assert torch.equal(rs, a)
    own.reduce_scatter(a[:], a, 0).wait()       # in place
    b = a.clone()
    own.broadcast(b, 0).wait()
    assert torch.equal(b, a)
    rows = torch.randn(37, 64, device="cuda")
    got = torch.zeros_like(rows)
    own.all_to_all(got, rows, [37], [37]).wait()
    assert torch.equal(got, rows)
    got.zero_()
    own.all_to_all(got, rows, [], []).synchronize()
    assert torch.equal(got, rows)
    for dt in (torch.int32, torch.int64):     # the expert-parallel dispatcher exchanges its row counts this way
        cnt = torch.arange(1, 9, device="cuda", dtype=dt)
        back = torch.zeros_like(cnt)
        own.all_to_all(back, cnt, [], []).wait()
        assert back.tolist() == list(range(1, 9)), (dt, back.tolist())
    empty = torch.empty(0, 64, device="cuda")
    own.all_to_all(torch.empty(0, 64, device="cuda"), empty, [0], [0]).wait()
    own.barrier()
    own.synchronize()
    assert own.num_calls() >= 9


def test_argument_checks_fire_on_the_host(own):
    a = torch.randn(64, device="cuda")
    with pytest.raises(RuntimeError):
        own.all_reduce(torch.randn(64), 0)                       # host tensor
    with pytest.raises(RuntimeError):
        own.all_reduce(torch.randn(8, 8, device="cuda").t(), 0)   # not contiguous
    with pytest.raises(RuntimeError):
        own.all_gather(torch.empty(100, device="cuda"), a)        # wrong size
    with pytest.raises(RuntimeError):
        own.reduce_scatter(torch.empty(32, device="cuda"), a, 0)
    with pytest.raises(RuntimeError):
        own.all_to_all(torch.empty(8, 4, device="cuda"), torch.empty(8, 4, device="cuda"), [5], [8])
    with pytest.raises(RuntimeError):
        own.broadcast(a, 3)
    with pytest.raises(RuntimeError):
        own.all_reduce(a, 9)
    with pytest.raises(RuntimeError):
        torch.classes.dpa.RcclComm(torch.zeros(5, dtype=torch.uint8), 0, 1, 0, 0, 0, 0)


def test_device_timings(own):
    own.set_timing(True)
    t = torch.randn(1 << 24, device="cuda").bfloat16()
    for _ in range(3):
        own.all_reduce(t, 0)
    own.synchronize()
    rec = own.drain_timings()
    own.set_timing(False)
    assert len(rec) == 3 and all(k == "all_reduce" and b == t.numel() * 2 and ms >= 0 for k, b, ms in rec)
    assert own.drain_timings() == []


@pytest.fixture()
def own_as_default_group(own):
    """A one-rank job whose GPU collectives all go through the own communicator (host-side objects over gloo)."""
    import torch.distributed as dist
    from distributed_pytorch_amd.parallel import comm
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29583", rank=0, world_size=1)
    old = comm.FORCE_COLLECTIVES, comm._OWN
    comm.FORCE_COLLECTIVES, comm._OWN = True, own
    yield comm.DistInfo(0, 0, 1, torch.device("cuda:0"), "rccl")
    comm.FORCE_COLLECTIVES, comm._OWN = old
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "zero2", "fsdp", "ep"])
def test_engines_over_the_own_communicator_match_single(own_as_default_group, own, mode):
    from distributed_pytorch_amd.config import LLMconfig, Trainconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.trainer import build_engine
    from test_dist_gpu import _run
    moe = dict(non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2) if mode == "ep" else dict(non_linearity="gelu")
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                    pos_emb="learn", **moe).finalize()
    tc = Trainconfig(dtype="bf16", learning_rate=1e-3, bucket_mb=0.5)
    torch.manual_seed(1)
    ref = _run(build_engine("single", LLM(cfg), tc, own_as_default_group), cfg)
    calls, before = own.num_calls(), dict(comm.STATS)
    import importlib
    lin = importlib.import_module("distributed_pytorch_amd.ops.linear")   # (ops.linear the attribute is the function)
    lin._Deferral.log.clear()
    torch.manual_seed(1)
    assert own.num_registered() == 0
    eng = build_engine(mode, LLM(cfg), tc, own_as_default_group)
    assert eng._dist
    # the persistent flat buffers of the engine were offered to RCCL for registration (it may decline them all)
    assert eng.registered_buffers == own.num_registered() <= len(eng._comm_buffers())
    assert len(eng._comm_buffers()) >= (2 if mode != "ddp" else 1)
    got = _run(eng, cfg)
    if mode == "ep":
        # the local experts' weight gradients were deferred behind the exchange of their input gradient and ran
        kinds = [k for k, _ in lin._Deferral.log]
        assert "defer" in kinds and "run" in kinds and not lin._Deferral.queue
    # registrations are released while the buffers are still allocated
    assert comm.deregister_buffers() == eng.registered_buffers and own.num_registered() == 0
    assert own.num_calls() > calls + 6, "the collectives did not go through the own communicator"
    moved = {k: comm.STATS[k] - before[k] for k in before}
    if mode in ("zero2", "fsdp"):
        assert moved["reduce_scatter"] > 0 and moved["all_gather"] > 0
    if mode == "ep":
        assert moved["all_to_all"] > 0
    assert own.async_error() == ""
    for a, b in zip(got, ref):
        assert abs(a - b) < 3e-2 * abs(b) + 1e-3, (mode, got, ref)


@pytest.mark.parametrize("mode", ["ddp", "zero2", "fsdp", "ep", "ep_fsdp"])
def test_engine_steps_over_the_own_communicator_hold_no_memory(own_as_default_group, own, mode):
    """Work handles of the own communicator keep their operands until the collective has finished; none of them (nor
    anything else of a step) may outlive the step: allocated memory is flat from step 4 to step 12 with the cyclic
    collector off."""
    import gc
    from distributed_pytorch_amd.config import LLMconfig, Trainconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.trainer import build_engine
    from test_dist_gpu import _run
    moe = dict(non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2) if mode.startswith("ep") else dict(non_linearity="gelu")
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                    pos_emb="learn", **moe).finalize()
    tc = Trainconfig(dtype="bf16", learning_rate=1e-3, bucket_mb=0.5)
    torch.manual_seed(1)
    eng = build_engine(mode, LLM(cfg), tc, own_as_default_group)
    gc.collect()
    gc.disable()
    try:
        _run(eng, cfg, steps=4)
        torch.cuda.synchronize()
        at4 = torch.cuda.memory_allocated()
        _run(eng, cfg, steps=8)
        torch.cuda.synchronize()
        at12 = torch.cuda.memory_allocated()
    finally:
        gc.enable()
        comm.deregister_buffers()
    assert at12 <= at4 + (1 << 20), f"allocated memory grew from {at4} to {at12} bytes over 8 steps"
    assert own.async_error() == ""


def test_buffer_registration(own):
    """A registered persistent buffer is used like any other; it is deregistered before it is freed."""
    t = torch.randn(1 << 22, device="cuda").bfloat16()
    want = t.clone()
    registered = own.register_buffer(t)        # the library may decline; the call must not fail
    own.all_reduce(t, 0).wait()
    torch.cuda.synchronize()
    assert torch.equal(t, want)
    assert own.deregister_buffers() == (1 if registered else 0)
    own.all_reduce(t, 0).wait()
    torch.cuda.synchronize()
    assert torch.equal(t, want)
