"""End-to-end checks on the MI355X: the bf16 HIP path against the fp32 eager path of the same
model, every engine type stepping on the GPU, and the HIP extension actually being used."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_pytorch_amd.config import LLMconfig
from distributed_pytorch_amd.models.llm import LLM

BASE = dict(vocab_size=1024, block_size=256, n_embd=256, n_layer=2, n_head=4, up_dim=512)
VARIANTS = {
    "gpt2": dict(attn="mha", pos_emb="learn", non_linearity="gelu"),
    "gqa_rope_swiglu": dict(attn="gqa", n_kv_heads=2, pos_emb="rope", non_linearity="swiglu"),
    "mqa_sin": dict(attn="mqa", pos_emb="sin", non_linearity="relu"),
    "mla_rope_flash": dict(attn="mla", pos_emb="rope", q_latent_dim=64, kv_latent_dim=96, rope_head_dim=32),
    "mla_naive_flash": dict(attn="mla", pos_emb="learn", q_latent_dim=64, kv_latent_dim=64),
    "mla_rope_small_latent": dict(attn="mla", pos_emb="rope", q_latent_dim=32, kv_latent_dim=32, rope_head_dim=16),
    # the reference's own launch configuration of latent attention (single-gpu/train.sh:34-36): 128-wide heads with
    # 128 rotary columns -> (key, value) widths (256, 128)
    "mla_rope_wide": dict(attn="mla", pos_emb="rope", n_embd=1024, n_head=8, up_dim=2048, q_latent_dim=256,
                          kv_latent_dim=256, rope_head_dim=128),
    "mla_rope_128_64": dict(attn="mla", pos_emb="rope", q_latent_dim=64, kv_latent_dim=64, rope_head_dim=64),
    "mha_hs32": dict(attn="mha", n_head=8, pos_emb="learn", non_linearity="gelu"),
    "moe": dict(attn="mha", pos_emb="learn", non_linearity="swiglu", moe=True, n_exp=4, n_shared=1, n_act=2),
    "recompute": dict(attn="mha", pos_emb="learn", non_linearity="gelu", act_recomp=True),
    "recompute_block": dict(attn="mha", pos_emb="learn", non_linearity="gelu", act_recomp=True,
                            act_recomp_scope="block"),
}


def test_extension_loaded():
    from distributed_pytorch_amd import _ext
    assert _ext.available(), "the in-tree HIP extension must load on the GPU box"
    assert hasattr(torch.ops.dpa, "attn_fwd") and hasattr(torch.ops.dpa, "wgrad2")


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_bf16_hip_matches_fp32_eager(name):
    from distributed_pytorch_amd.parallel.engine import Engine
    torch.manual_seed(0)
    cfg = LLMconfig(**{**BASE, **VARIANTS[name]}).finalize()
    ref = LLM(cfg)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    idx = torch.randint(0, cfg.vocab_size, (4, 256))
    tgt = torch.randint(0, cfg.vocab_size, (4, 256))
    _, want, _ = ref(idx, tgt)
    want.backward()
    gref = {n: p.grad.clone() for n, p in ref.named_parameters()}
    torch.manual_seed(0)
    model = LLM(cfg)
    model.load_state_dict(sd)
    eng = Engine(model, dtype="bf16", device="cuda:0")
    eng.zero_grad()
    _, got, _ = eng(idx.cuda(), tgt.cuda())
    got.backward()
    assert abs(got.item() - want.item()) < 3e-2 * abs(want.item())
    # every big gradient tensor agrees with the fp32 model to bf16 accuracy: relative L2 error per tensor.
    # Models with a discontinuity in the forward pass get a wider bound: bf16 rounding flips the ReLU mask
    # of the ~0.3 % of pre-activations that sit within rounding distance of zero, and each flip is a
    # full-size error (sqrt(0.003) ~ 5 %); the MoE router's top-k choice flips the same way.
    tol = {"mqa_sin": 8e-2, "moe": 2e-1}.get(name, 3e-2)
    worst = (0.0, "")
    for n, p in eng.module.named_parameters():
        if p.numel() >= 4096 and "gate" not in n:
            a, b = p.grad.float().flatten().cpu(), gref[n].flatten()
            err = ((a - b).norm() / (b.norm() + 1e-12)).item()
            worst = max(worst, (err, n))
            # a routed expert's gradient is a sum over the few tokens routed to it: one flipped top-k choice
            # moves a whole token's contribution in or out
            bound = 0.35 if (name == "moe" and ".experts." in n) else tol
            assert err < bound, f"{name}: grad of {n} is off by {err:.4f} (relative L2)"
    print(f"{name}: worst relative gradient error {worst[0]:.4f} ({worst[1]})")


@pytest.mark.parametrize("mode", ["single", "ddp", "zero1", "zero2", "fsdp"])
def test_every_engine_trains_on_gpu(mode):
    from distributed_pytorch_amd.config import Trainconfig
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.trainer import build_engine
    torch.manual_seed(1)
    cfg = LLMconfig(**BASE, **VARIANTS["gpt2"]).finalize()
    info = comm.DistInfo(0, 0, 1, torch.device("cuda:0"), "none")
    eng = build_engine(mode, LLM(cfg), Trainconfig(dtype="bf16", learning_rate=1e-3), info)
    x = torch.randint(0, cfg.vocab_size, (8, 256), device="cuda")
    losses = []
    for it in range(8):
        eng.zero_grad()
        for micro in range(2):
            eng.require_backward_grad_sync = micro == 1
            _, loss, _ = eng(x, x.roll(-1, 1))
            (loss / 2).backward()
        eng.clip_grad_norm_(1.0)
        eng.step(3e-3)
        losses.append(loss.item())
    assert losses[-1] < losses[0] - 0.5, f"{mode}: {losses}"
    sd = eng.state_dict()
    assert sd["lm_head.weight"].dtype == torch.float32 and sd["lm_head.weight"].shape == (1024, 256)


@pytest.mark.parametrize("name", ["gqa_rope_swiglu", "mla_rope_wide"])
def test_generate_on_gpu_matches_full_forward(name):
    """Incremental decoding over the KV cache reproduces the full forward pass. The wide latent configuration has no
    static-cache decode path (its absorbed key is 384 wide): its cache path runs the absorbed form eagerly, while its
    cache-free forward in evaluation mode is the absorbed form too, so the two must agree."""
    torch.manual_seed(2)
    cfg = LLMconfig(**{**BASE, **VARIANTS[name]}).finalize()
    m = LLM(cfg).cuda().to(torch.bfloat16).eval()
    assert m.rope_cos.dtype == torch.float32 and m.freqs_cis.is_complex()   # tables survive the cast
    idx = torch.randint(0, cfg.vocab_size, (2, 40), device="cuda")
    with torch.no_grad():
        logits, _, caches = m(idx[:, :32])
        for t in range(32, 40):
            logits, _, caches = m(idx[:, t:t + 1], kv_caches=caches)
        full, _, _ = m(idx)
    assert torch.allclose(logits[:, -1].float(), full[:, -1].float(), rtol=5e-2, atol=5e-2)
    out = m.generate(idx[:, :8], max_new_tokens=16, topk=1)
    assert out.shape == (2, 24)


def test_graphed_step_matches_eager_steps():
    """The optimizer step replayed from one hipGraph follows the eager trajectory (same kernels,
    lr schedule and Adam bias corrections fed from device memory)."""
    from distributed_pytorch_amd.parallel.engine import Engine
    from distributed_pytorch_amd.parallel.graph import GraphedTrainStep
    cfg = LLMconfig(**BASE, **VARIANTS["gpt2"]).finalize()
    torch.manual_seed(5)
    data = [torch.randint(0, cfg.vocab_size, (2, 4, 256), device="cuda") for _ in range(10)]
    lrs = [1e-3 * (i + 1) / 4 if i < 4 else 1e-3 * 0.9 ** i for i in range(10)]

    def run(graph: bool):
        torch.manual_seed(1)
        eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3)
        gs = GraphedTrainStep(eng, accum=2, grad_clip=1.0, warmup=2) if graph else None
        out = []
        for xb, lr in zip(data, lrs):
            batches = [(xb[i], xb[i].roll(-1, 1)) for i in range(2)]
            if graph:
                out.append(gs.step(batches, lr).item())
            else:
                eng.zero_grad()
                for bx, by in batches:
                    _, loss, _ = eng(bx, by)
                    (loss / 2).backward()
                eng.clip_grad_norm_(1.0)
                eng.step(lr)
                out.append(loss.item())
        return out, eng

    eager, e1 = run(False)
    graphed, e2 = run(True)
    assert e1.step_count == e2.step_count == 10
    for a, b in zip(graphed, eager):
        assert abs(a - b) < 2e-2 * abs(b), (graphed, eager)
    w1 = e1.state_dict()["lm_head.weight"]
    w2 = e2.state_dict()["lm_head.weight"]
    assert ((w1 - w2).norm() / w1.norm()).item() < 2e-2


def test_graphed_step_with_dropout_captures_and_redraws_masks():
    """dropout > 0 under hipGraph replay: the seed draw is a generator op the capture registers, so
    the step is captured (no eager fallback) and replays of the SAME batch give different losses."""
    from distributed_pytorch_amd.parallel.engine import Engine
    from distributed_pytorch_amd.parallel.graph import GraphedTrainStep
    cfg = LLMconfig(**{**BASE, **VARIANTS["gpt2"], "dropout": 0.2}).finalize()
    torch.manual_seed(2)
    eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0", lr=0.0)
    eng.train()
    gs = GraphedTrainStep(eng, accum=1, grad_clip=1.0, warmup=1)
    x = torch.randint(0, cfg.vocab_size, (4, 256), device="cuda")
    losses = [gs.step([(x, x.roll(-1, 1))], 0.0).item() for _ in range(5)]
    assert gs.graph is not None and not gs.failed
    assert len({round(v, 6) for v in losses[2:]}) > 1, losses   # lr = 0: only the masks change


def test_graphed_step_falls_back_to_eager_when_the_capture_fails():
    """A step that cannot be captured (here: a host read in the middle of it) is reported once and training
    continues eagerly, with the runtime's sticky capture error cleared."""
    from distributed_pytorch_amd.parallel.engine import Engine
    from distributed_pytorch_amd.parallel.graph import GraphedTrainStep
    cfg = LLMconfig(**{**BASE, **VARIANTS["gpt2"]}).finalize()
    torch.manual_seed(3)
    eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3)
    eng.train()
    gs = GraphedTrainStep(eng, accum=1, grad_clip=1.0, warmup=1)
    body = gs._body

    def body_with_host_read(xs, ys):
        loss = body(xs, ys)
        if torch.cuda.is_current_stream_capturing():
            loss.item()                       # a synchronising read: not allowed while capturing
        return loss
    gs._body = body_with_host_read
    x = torch.randint(0, cfg.vocab_size, (4, 256), device="cuda")
    losses = [gs.step([(x, x.roll(-1, 1))], 1e-3).item() for _ in range(5)]
    assert gs.failed and gs.graph is None
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]


def test_graphed_step_captures_moe_with_swiglu_experts():
    """Experts on the grouped NT GEMM under hipGraph capture: the per-step transposed weights keep the addresses of
    the warm-up steps, so the cached pointer tables stay valid and nothing is copied from the host while capturing."""
    from distributed_pytorch_amd.parallel.engine import Engine
    from distributed_pytorch_amd.parallel.graph import GraphedTrainStep
    cfg = LLMconfig(**{**BASE, "attn": "mha", "pos_emb": "learn", "non_linearity": "swiglu", "moe": True, "n_exp": 4,
                       "n_shared": 1, "n_act": 2}).finalize()
    torch.manual_seed(4)
    eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0", lr=1e-3)
    eng.train()
    gs = GraphedTrainStep(eng, accum=1, grad_clip=1.0, warmup=2)
    x = torch.randint(0, cfg.vocab_size, (4, 256), device="cuda")
    losses = [gs.step([(x, x.roll(-1, 1))], 1e-3).item() for _ in range(8)]
    assert gs.graph is not None and not gs.failed
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]


def test_trainer_resume_on_gpu(tmp_path):
    """Save a resume file mid-run on the GPU, continue from it in a new trainer call (loader RNG state
    comes back from a file mapped to the GPU) and land on the same loss as the uninterrupted run."""
    from distributed_pytorch_amd.config import Trainconfig
    from distributed_pytorch_amd.trainer import train
    mc = LLMconfig(**BASE, **VARIANTS["gpt2"]).finalize()

    def tc(max_iters, resume=""):
        return Trainconfig(dataset="synthetic", synthetic_tokens=200_000, data_dir=str(tmp_path / "data"), batch_size=4,
                           total_batch_size=4 * 256, max_iters=max_iters, learning_rate=1e-3, warmup_steps=2,
                           dtype="bf16", save_interval=4, file_name=str(tmp_path / "run"), resume=resume, log_interval=100)
    full = train(tc(8), mc, "single")
    train(tc(4), mc, "single")                       # writes run_resume.rank0 at iteration 4
    cont = train(tc(8, resume=str(tmp_path / "run_resume")), mc, "single")
    assert abs(cont["final_loss"] - full["final_loss"]) < 2e-2 * abs(full["final_loss"])


@pytest.mark.parametrize("scope", ["none", "attn", "block"])
def test_dropout_trains_on_the_hip_path(scope):
    """dropout > 0 stays on the flash kernels (in-kernel mask); with activation recomputation the
    recomputed forward must redraw the same mask, which shows as gradients equal to the plain run."""
    from distributed_pytorch_amd.parallel.engine import Engine
    grads = {}
    for sc in ("none", scope):
        torch.manual_seed(0)
        cfg = LLMconfig(**{**BASE, **VARIANTS["gpt2"], "dropout": 0.1, "act_recomp": sc != "none",
                           "act_recomp_scope": sc if sc != "none" else "attn"}).finalize()
        eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0")
        eng.train()
        idx = torch.randint(0, cfg.vocab_size, (4, 256), device="cuda")
        torch.manual_seed(11)
        eng.zero_grad()
        _, loss, _ = eng(idx, idx.roll(-1, 1))
        loss.backward()
        assert torch.isfinite(loss)
        grads[sc] = {n: p.grad.float().clone() for n, p in eng.module.named_parameters()}
    for n, g in grads["none"].items():
        other = grads[scope][n]
        assert (g - other).norm() <= 2e-2 * g.norm() + 1e-6, n


@pytest.mark.parametrize("name", ["gpt2", "gqa_rope_swiglu", "mqa_sin", "moe", "mla_rope_flash", "mla_naive_flash",
                                  "mla_rope_small_latent"])
def test_graphed_decode_matches_eager_decode(name):
    """generate() replaying one captured decode step per token (static KV cache, device-side positions)
    produces the tokens of the eager cache path; top-k 1 makes sampling deterministic."""
    torch.manual_seed(0)
    cfg = LLMconfig(**{**BASE, **VARIANTS[name]}).finalize()
    model = LLM(cfg).to("cuda").to(torch.bfloat16)
    model.eval()
    idx = torch.randint(0, cfg.vocab_size, (3, 17), device="cuda")
    eager = model.generate(idx, 40, temperature=1.0, topk=1, hip_graph=False)
    graphed = model.generate(idx, 40, temperature=1.0, topk=1, hip_graph=True)
    assert model.last_decode_graphed, "the decode step must be captured, not silently run eagerly"
    assert graphed.shape == eager.shape == (3, 57)
    assert torch.equal(graphed[:, :17], idx)
    agree = (graphed == eager).float().mean().item()
    assert agree > 0.97, agree            # identical kernels and maths; a bf16 near-tie may flip an argmax
    sampled = model.generate(idx, 40, temperature=0.8, topk=50)       # default: graphed, real sampling
    assert sampled.shape == (3, 57) and int(sampled.max()) < cfg.vocab_size


@pytest.mark.parametrize("batch", [1, 16, 17, 64, 65])
def test_graphed_decode_batch_sizes_around_the_fused_kernels_limit(batch):
    """The captured decode step on the one-launch decode layers (up to 64 sequences, in groups of 16 rows) and on
    the stock launches (65: beyond the kernels' rows) produces the eager cache path's tokens."""
    torch.manual_seed(0)
    cfg = LLMconfig(**{**BASE, **VARIANTS["gpt2"]}).finalize()
    model = LLM(cfg).to("cuda").to(torch.bfloat16)
    model.eval()
    idx = torch.randint(0, cfg.vocab_size, (batch, 9), device="cuda")
    eager = model.generate(idx, 24, temperature=1.0, topk=1, hip_graph=False)
    graphed = model.generate(idx, 24, temperature=1.0, topk=1, hip_graph=True)
    assert model.last_decode_graphed
    assert graphed.shape == eager.shape == (batch, 33)
    assert (graphed == eager).float().mean().item() > 0.97
    x = torch.randn(batch, 1, cfg.n_embd, device="cuda").bfloat16()
    assert model.transformer.h[0]._decode_ok(x) == (batch <= 64)


def test_rope_block_trains_when_the_projection_runs_on_the_library(monkeypatch):
    """A rotary model whose qkv projection is below the own GEMM's dispatch bar: the library product's output is
    rotated in place like the own kernel's (it must not be a tracked view of a custom Function's result)."""
    import importlib
    L = importlib.import_module("distributed_pytorch_amd.ops.linear")
    monkeypatch.setattr(L, "GEMM_MIN_TILES", 1 << 20)          # every product to the library
    torch.manual_seed(0)
    cfg = LLMconfig(**{**BASE, **VARIANTS["gqa_rope_swiglu"]}).finalize()
    model = LLM(cfg).to("cuda").to(torch.bfloat16)
    idx = torch.randint(0, cfg.vocab_size, (4, cfg.block_size), device="cuda")
    _, loss, _ = model(idx, idx)
    loss.backward()
    g = model.transformer.h[0].attn.attn.c_attn.weight.grad
    assert torch.isfinite(loss) and g is not None and torch.isfinite(g).all() and g.abs().max() > 0


def test_moe_with_dropout_stays_on_the_grouped_kernels_and_trains():
    """A MoE model with dropout = 0.1: the routed experts keep running on the grouped kernels (no per-expert loop, no
    host read of the routing: the step can be captured), the loss is finite and falls over a few steps."""
    from distributed_pytorch_amd.parallel.engine import Engine
    from distributed_pytorch_amd import ops
    torch.manual_seed(3)
    cfg = LLMconfig(**{**BASE, **VARIANTS["moe"], "dropout": 0.1}).finalize()
    eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0")
    opt = eng.configure_optimizers(learning_rate=3e-3)
    idx = torch.randint(0, cfg.vocab_size, (4, 256), device="cuda")
    tgt = torch.randint(0, cfg.vocab_size, (4, 256), device="cuda")
    calls = {"n": 0}
    real = ops.expert_counts

    def counting(*a, **kw):
        calls["n"] += 1
        return real(*a, **kw)

    ops.expert_counts = counting     # the per-expert fallback reads the counts back to the host through this
    try:
        losses = []
        for _ in range(8):
            eng.zero_grad()
            _, loss, _ = eng(idx, tgt)
            loss.backward()
            eng.clip_grad_norm_(1.0)
            opt.step()
            losses.append(loss.item())
    finally:
        ops.expert_counts = real
    assert calls["n"] == 0, "the routed experts fell back to the per-expert loop"
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
    # two forward passes in training mode differ (dropout is live), two in eval mode do not
    eng.module.train()
    a = eng(idx, tgt)[1].item()
    b = eng(idx, tgt)[1].item()
    eng.module.eval()
    c = eng(idx, tgt)[1].item()
    d = eng(idx, tgt)[1].item()
    assert a != b and c == d


@pytest.mark.parametrize("name", ["gpt2", "gqa_rope_swiglu"])
def test_bf16_hip_matches_fp32_eager_at_gpt2_width(name):
    """The same comparison at the width, sequence length and head count of the benchmarked model (C = 768, T = 1024,
    12 heads; two layers, batch 2): the shapes the tuned kernel paths (256-row GEMM tiles, fused projection + rotary
    + attention, 64-wide heads at 8 key tiles per query tile) actually run at."""
    from distributed_pytorch_amd.parallel.engine import Engine
    torch.manual_seed(0)
    over = dict(vocab_size=2048, block_size=1024, n_embd=768, n_layer=2, n_head=12, up_dim=3072)
    if name == "gqa_rope_swiglu":
        over["n_kv_heads"] = 4
    cfg = LLMconfig(**{**BASE, **VARIANTS[name], **over}).finalize()
    ref = LLM(cfg)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    idx = torch.randint(0, cfg.vocab_size, (2, 1024))
    tgt = torch.randint(0, cfg.vocab_size, (2, 1024))
    _, want, _ = ref(idx, tgt)
    want.backward()
    gref = {n: p.grad.clone() for n, p in ref.named_parameters()}
    model = LLM(cfg)
    model.load_state_dict(sd)
    eng = Engine(model, dtype="bf16", device="cuda:0")
    eng.zero_grad()
    _, got, _ = eng(idx.cuda(), tgt.cuda())
    got.backward()
    assert abs(got.item() - want.item()) < 3e-2 * abs(want.item())
    for n, p in eng.module.named_parameters():
        if p.numel() >= 4096:
            a, b = p.grad.float().flatten().cpu(), gref[n].flatten()
            err = ((a - b).norm() / (b.norm() + 1e-12)).item()
            assert err < 3e-2, f"{name}: grad of {n} is off by {err:.4f} (relative L2)"


def test_rope_fallback_backward_keeps_the_incoming_gradient_intact():
    """The in-place rotary op outside the fused attention path (eager attention, here forced): backward twice over a
    retained graph gives the same gradients both times, i.e. the op does not rotate the gradient buffer it is handed."""
    from distributed_pytorch_amd.ops.rope import rope_qkv_, rope_tables
    torch.manual_seed(5)
    B, T, nh, nkv, hs = 2, 128, 4, 2, 64
    W = (nh + 2 * nkv) * hs
    cos_t, sin_t = rope_tables(256, hs, device="cuda")
    x = torch.randn(B, T, W, device="cuda").bfloat16().requires_grad_()
    w = torch.randn(W, device="cuda").bfloat16()
    y = rope_qkv_(x * 1.0, cos_t, sin_t, 0, nh + nkv, hs)
    loss = (y.float() * w.float()).sum()
    loss.backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    loss.backward()
    assert torch.equal(g1, x.grad)


@pytest.mark.parametrize("name", ["gpt2", "gqa_rope_swiglu"])
def test_training_trajectory_bf16_hip_follows_fp32_cpu_engine(name):
    """40 optimizer steps on a passage the model can learn: the bf16 engine on the HIP kernels (fused AdamW, fused
    clip, fused head + loss) against the fp32 engine on the CPU, whose trajectory tests/test_reference_trajectory_cpu.py
    pins to the reference model under the reference's recipe. Same initial weights, batches and learning rates."""
    from distributed_pytorch_amd.config import Trainconfig
    from distributed_pytorch_amd.parallel import comm
    from distributed_pytorch_amd.trainer import build_engine
    torch.manual_seed(5)
    cfg = LLMconfig(**{**BASE, **VARIANTS[name]}).finalize()
    first = LLM(cfg)
    sd = {k: v.clone() for k, v in first.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    corpus = torch.randint(0, cfg.vocab_size, (1500,), generator=g).repeat(6)
    B, T, steps = 4, 256, 40

    def batch(it):
        gb = torch.Generator().manual_seed(100 + it)
        at = torch.randint(0, corpus.numel() - T - 1, (B,), generator=gb)
        return torch.stack([corpus[a:a + T] for a in at]), torch.stack([corpus[a + 1:a + T + 1] for a in at])

    def run(device, dtype):
        model = LLM(cfg)
        model.load_state_dict(sd)
        tc = Trainconfig(dtype=dtype, grad_dtype=dtype, learning_rate=2e-3, weight_decay=0.1, grad_clip=1.0)
        eng = build_engine("single", model, tc, comm.DistInfo(0, 0, 1, torch.device(device), "none"))
        out = []
        for it in range(steps):
            x, y = batch(it)
            eng.zero_grad()
            _, loss, _ = eng(x.to(device), y.to(device))
            loss.backward()
            eng.clip_grad_norm_(1.0)
            eng.step(2e-3 * min(1.0, (it + 1) / 5))
            out.append(loss.item())
        return out

    want = run("cpu", "fp32")
    got = run("cuda:0", "bf16")
    assert want[-1] < 0.6 * want[0], f"the fp32 run has to learn: {want[0]:.3f} -> {want[-1]:.3f}"
    for s in range(0, steps, 5):
        a, b = sum(want[s:s + 5]) / 5, sum(got[s:s + 5]) / 5
        assert abs(a - b) < 0.05 * a + 0.02, f"{name}: steps {s}-{s + 4}: fp32 {a:.4f}, bf16 HIP {b:.4f}\n{want}\n{got}"


@pytest.mark.gpu
def test_background_resume_snapshot_on_the_gpu(tmp_path):
    """--save_async on the GPU: the snapshot goes through pinned staging buffers (reused by the next save), the writer
    thread serialises it while the optimizer moves on, and what loads back is the state at the call."""
    from distributed_pytorch_amd import checkpoint as ck
    from distributed_pytorch_amd.config import LLMconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel.engine import Engine
    cfg = LLMconfig(vocab_size=1024, block_size=128, n_embd=256, n_layer=2, n_head=4, up_dim=512, attn="mha", pos_emb="learn",
                    non_linearity="gelu").finalize()

    def make():
        torch.manual_seed(0)
        return Engine(LLM(cfg), device="cuda:0", dtype="bf16")

    def step(eng, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randint(0, cfg.vocab_size, (4, 128), generator=g).cuda()
        eng.zero_grad()
        _, loss, _ = eng(x, x.roll(-1, 1))
        loss.backward()
        eng.clip_grad_norm_(1.0)
        eng.step(1e-3)

    path = str(tmp_path / "run_resume")
    eng = make()
    truth = {}
    for it in range(3):
        step(eng, it)
        truth[it] = [(o.master.clone(), o.exp_avg_sq.clone()) for o in eng.optims]
        ck.save_resume(path, eng, it, (), 0, 1, background=True)
        staged = {k: v.data_ptr() for k, v in ck._PINNED.items()}
        if it:
            assert staged == first_staged           # the staging buffers of the first snapshot serve every later one
        first_staged = staged
    step(eng, 3)                                    # the live state moves past the last snapshot before its commit
    assert ck.finish_pending_save() == 2
    fresh = make()
    assert ck.load_resume(path, fresh, (), 0) == 2
    for o, (m, v) in zip(fresh.optims, truth[2]):
        assert torch.equal(o.master, m) and torch.equal(o.exp_avg_sq, v)


def test_phase_timer_holds_a_bounded_number_of_events():
    """Event mode of the profiler over a long run: finished event pairs are folded into the totals as the run goes."""
    from distributed_pytorch_amd.utils.profiling import PhaseTimer
    t = PhaseTimer("events")
    x = torch.ones(1024, device="cuda")
    for _ in range(2000):
        with t.phase("forward"):
            x = x * 1.0001
    assert len(t._pending) < 600 and t.cnt["forward"] == 2000
    rep = t.report()
    assert "forward" in rep and not t._pending
    assert t.tot["forward"] > 0
