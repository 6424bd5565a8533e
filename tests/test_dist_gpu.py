"""RCCL call paths on one GPU: a one-rank ``nccl`` process group with every collective really
issued (``comm.FORCE_COLLECTIVES``), so the in-place reduce-scatter / all-gather views, the async
work handles and the stream ordering of the engines are exercised exactly as a multi-rank job
issues them; and the ``bench.py`` launch line of the multi-GPU runs with one rank."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist
    from distributed_pytorch_amd.parallel import comm
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    old = comm.FORCE_COLLECTIVES
    comm.FORCE_COLLECTIVES = True
    yield comm.DistInfo(0, 0, 1, torch.device("cuda:0"), "nccl")
    comm.FORCE_COLLECTIVES = old
    dist.destroy_process_group()


def _run(engine, cfg, steps=6):
    torch.manual_seed(3)
    x = torch.randint(0, cfg.vocab_size, (8, 256), device="cuda")
    losses = []
    for _ in range(steps):
        engine.zero_grad()
        for micro in range(2):
            engine.require_backward_grad_sync = micro == 1
            _, loss, _ = engine(x, x.roll(-1, 1))
            (loss / 2).backward()
        engine.clip_grad_norm_(1.0)
        engine.step(3e-3)
        losses.append(loss.item())
    return losses


@pytest.mark.parametrize("mode", ["ddp", "zero1", "zero2", "fsdp", "ddp_vp", "fsdp_vp"])
def test_engine_through_rccl_matches_single(one_rank_group, mode):
    from distributed_pytorch_amd.config import LLMconfig, Trainconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.trainer import build_engine
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                    pos_emb="learn", non_linearity="gelu").finalize()
    tc = Trainconfig(dtype="bf16", learning_rate=1e-3, bucket_mb=0.5)
    torch.manual_seed(1)
    ref = _run(build_engine("single", LLM(cfg), tc, one_rank_group), cfg)
    before = dict(comm.STATS)
    torch.manual_seed(1)
    if mode.endswith("_vp"):      # head + loss over the vocabulary-parallel kernels (stats / all-reduce / grad)
        tc.vocab_parallel, mode = True, mode[:-3]
    eng = build_engine(mode, LLM(cfg), tc, one_rank_group)
    assert (eng.module.vocab_parallel is not None) == tc.vocab_parallel
    assert eng._dist, "collectives must really be issued in this test"
    got = _run(eng, cfg)
    moved = {k: comm.STATS[k] - before[k] for k in before}
    assert sum(moved.values()) > 0, "no collective was issued"
    if mode == "zero2":
        assert moved["reduce_scatter"] > 0 and moved["all_gather"] > 0
    if mode == "fsdp":
        assert moved["reduce_scatter"] > 0 and moved["all_gather"] > 0
    for a, b in zip(got, ref):
        assert abs(a - b) < 2e-2 * abs(b) + 1e-3, (mode, got, ref)


@pytest.mark.parametrize("kind,capacity", [("ep", 0.0), ("ep_fsdp", 0.0), ("ep", 1.0)])
def test_expert_parallel_through_rccl(one_rank_group, kind, capacity):
    from distributed_pytorch_amd.config import LLMconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.parallel.engine import Engine
    from distributed_pytorch_amd.parallel.ep import EPEngine, EPFSDPEngine
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                    pos_emb="learn", non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2).finalize()
    torch.manual_seed(1)
    ref = _run(Engine(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3), cfg)
    before = comm.STATS["all_to_all"]
    torch.manual_seed(1)
    cls = EPEngine if kind == "ep" else EPFSDPEngine
    eng = cls(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3, ep_capacity=capacity)
    got = _run(eng, cfg)
    assert eng.dispatcher.dropped_pairs() == 0
    assert comm.STATS["all_to_all"] > before
    for a, b in zip(got, ref):
        assert abs(a - b) < 3e-2 * abs(b) + 1e-3, (got, ref)


@pytest.mark.parametrize("kind,capacity", [("ep", 0.0), ("ep_fsdp", 0.0), ("ep", 1.0)])
def test_expert_parallel_steps_hold_no_memory(one_rank_group, kind, capacity):
    """Nothing of a step survives it: with the cyclic garbage collector switched off (so that only reference
    counting frees), the allocated GPU memory after step 12 equals that after step 4. (The dispatcher's per-layer
    state once sat in a reference cycle with its own autograd nodes, which no collector walks: a training run
    grew by gigabytes per step until it ran out of memory.)"""
    import gc
    from distributed_pytorch_amd.config import LLMconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel.ep import EPEngine, EPFSDPEngine
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                    pos_emb="learn", non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2).finalize()
    torch.manual_seed(1)
    cls = EPEngine if kind == "ep" else EPFSDPEngine
    eng = cls(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3, ep_capacity=capacity)
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
    assert at12 <= at4 + (1 << 20), f"allocated memory grew from {at4} to {at12} bytes over 8 steps"


@pytest.mark.parametrize("mode", ["single", "ddp", "zero2", "fsdp", "fsdp_vp", "single_moe", "ddp_split", "single_block",
                                  "fsdp_block"])
def test_engine_steps_hold_no_memory(one_rank_group, mode):
    """The same for the dense engines and the local MoE path: allocated memory is flat from step 4 to step 12 with the
    cyclic collector off."""
    import gc
    from distributed_pytorch_amd.config import LLMconfig, Trainconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.trainer import build_engine
    moe = mode.endswith("_moe")
    block = mode.endswith("_block")     # whole-block recomputation
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="gqa", n_kv_heads=2,
                    pos_emb="rope", non_linearity="swiglu", moe=moe, n_exp=4, n_shared=1, n_act=2, act_recomp=block,
                    act_recomp_scope="block").finalize()
    tc = Trainconfig(learning_rate=1e-3, bucket_mb=0.5, dtype="bf16")
    if mode.endswith("_vp"):
        tc.vocab_parallel, mode = True, mode[:-3]
    if mode.endswith("_split"):         # the tied head weight reaches the loss as a fresh view tensor every step
        tc.tied_tail, mode = "split", mode[:-6]
    torch.manual_seed(1)
    eng = build_engine(mode.replace("_moe", "").replace("_block", ""), LLM(cfg), tc, one_rank_group)
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
    assert at12 <= at4 + (1 << 20), f"allocated memory grew from {at4} to {at12} bytes over 8 steps"


def test_vocab_parallel_cross_entropy_hip_halves(one_rank_group):
    """STATS + all-reduce + GRAD kernels (the two halves of vocab-parallel cross-entropy) against
    F.cross_entropy, with a vocabulary offset so that some targets fall outside the local shard."""
    import torch.nn.functional as F
    from distributed_pytorch_amd.ops import vocab_parallel_cross_entropy
    torch.manual_seed(0)
    N, V = 512, 4096
    logits = (torch.randn(N, V, device="cuda") * 3).to(torch.bfloat16)
    targets = torch.randint(0, V, (N,), device="cuda")
    targets[::7] = -1
    ref_in = logits.float().requires_grad_()
    ref = F.cross_entropy(ref_in, targets, ignore_index=-1)
    ref.backward()
    local = logits.clone().requires_grad_()
    loss = vocab_parallel_cross_entropy(local, targets, 0)
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-3 * abs(ref.item())
    err = (local.grad.float() - ref_in.grad).norm() / ref_in.grad.norm()
    assert err.item() < 1e-2


@pytest.mark.parametrize("mode,backend", [("ddp", "c10d"), ("fsdp", "c10d"), ("ddp", "own"), ("zero2", "own")])
def test_bench_launch_line_prints_one_json_line(mode, backend):
    """The multi-GPU launch line with one rank: stdout is exactly the JSON result, with the pinned
    protocol's statistics and the fields a scaling record is cross-checked against."""
    env = dict(os.environ, DPA_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29579", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--batch", "4", "--mode", mode, "--comm_backend", backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "baseline_source", "step_ms", "tokens_per_sec_median_step",
                "rccl_ranks", "comm_bytes_per_step", "n_buckets", "grad_dtype", "rccl_env"):
        assert key in out
    assert set(out["step_ms"]) == {"median", "mean", "p10", "p90"}
    assert out["step_ms"]["p10"] <= out["step_ms"]["median"] <= out["step_ms"]["p90"]
    assert out["rccl_ranks"] == 1 and out["n_buckets"] >= 1 and sum(out["comm_bytes_per_step"].values()) > 0
    assert out["rccl_env"]["backend"] == ("rccl(own)" if backend == "own" else "nccl")
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["seq_len"] == 1024 and out["config"]["global_batch"] == 4


def _two_rank_bench(mode, port, extra=()):
    env = dict(os.environ, DPA_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", DPA_HANG_DUMP_S="80")
    env.pop("DPA_FORCE_DIST", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--batch", "4", "--seq", "256", "--model", "char-10m", "--mode", mode, *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=140)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu_every_engine_agrees():
    """Two ranks on ONE GPU (gloo transport staged through the host, because RCCL refuses two ranks
    on a device): the GPU code paths of every engine run with a real peer (bucket hooks, shard
    maps, gathers of foreign shards) and all engines reach the same loss after the same steps."""
    losses = {}
    for i, mode in enumerate(["ddp", "zero1", "zero2", "fsdp"]):
        out = _two_rank_bench(mode, 29590 + i)
        assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8
        losses[mode] = out["final_loss"]
    ref = losses["ddp"]
    for mode, v in losses.items():
        assert abs(v - ref) < 5e-3 * abs(ref), losses


def test_graphed_ddp_step_captures_the_all_reduces(one_rank_group):
    """The data-parallel step, bucketed all-reduces included, replayed from one hipGraph follows the
    eager DDP trajectory."""
    from distributed_pytorch_amd.config import LLMconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel.ddp import DDPEngine
    from distributed_pytorch_amd.parallel.graph import GraphedTrainStep
    cfg = LLMconfig(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha",
                    pos_emb="learn", non_linearity="gelu").finalize()
    torch.manual_seed(5)
    data = [torch.randint(0, cfg.vocab_size, (2, 4, 256), device="cuda") for _ in range(8)]

    def run(graph):
        torch.manual_seed(1)
        eng = DDPEngine(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3, bucket_mb=0.5)
        gs = GraphedTrainStep(eng, accum=2, grad_clip=1.0, warmup=2) if graph else None
        out = []
        for xb in data:
            batches = [(xb[i], xb[i].roll(-1, 1)) for i in range(2)]
            if graph:
                out.append(gs.step(batches, 1e-3).item())
            else:
                eng.zero_grad()
                for i, (bx, by) in enumerate(batches):
                    eng.require_backward_grad_sync = i == 1
                    _, loss, _ = eng(bx, by)
                    (loss / 2).backward()
                eng.clip_grad_norm_(1.0)
                eng.step(1e-3)
                out.append(loss.item())
        return out, gs

    eager, _ = run(False)
    graphed, gs = run(True)
    assert gs.graph is not None and not gs.failed, "the DDP step must be captured"
    for a, b in zip(graphed, eager):
        assert abs(a - b) < 2e-2 * abs(b), (graphed, eager)
