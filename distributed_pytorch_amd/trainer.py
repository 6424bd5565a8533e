"""The one training loop behind every entry point (single GPU, DDP, ZeRO-1/2, FSDP).

Loop structure, LR schedule, evaluation and log lines follow the reference
(reference: single-gpu/train.py:263-372, multi-gpu/ddp/train.py:270-342,
multi-gpu/fsdp/kaggle-fsdp.py:1098-1139). Differences:

* bf16 compute with fp32 master weights and no gradient scaler (the reference's multi-GPU
  scripts hard-code fp16 + GradScaler); non-finite gradients skip the step inside the fused
  optimizer;
* the loss is read back to the host every ``sync_loss_every`` steps instead of every micro-step;
* each step line also carries tokens/s and MFU; ``--metrics_jsonl`` mirrors it to a file;
* the CPU path works (the reference's CPU path crashes on ``scaler.scale``);
* data shards are deterministic per rank; rank 0 alone prepares synthetic data;
* ``--save_model`` works for every strategy (state gathered to rank 0 in the reference key
  schema); ``--save_interval`` / ``--resume`` add restartable training.
"""
from __future__ import annotations

import math
import os
from time import perf_counter

import torch

from .checkpoint import finish_pending_save, load_resume, poll_pending_save, save_reference_checkpoint, save_resume
from .config import LLMconfig, Trainconfig
from .data import DataLoader, check_vocab, ensure_synthetic, resolve_data_dir
from .models.llm import LLM
from .parallel import comm
from .utils.metrics import JsonlSink, PEAK_BF16_DENSE_FLOPS, mfu, model_flops_per_token
from .utils.profiling import PhaseTimer


def get_lr(it: int, tc: Trainconfig) -> float:
    """Linear warm-up then cosine decay to 10 % (reference: single-gpu/train.py:263-278)."""
    max_lr = tc.learning_rate
    min_lr = max_lr * 0.1
    max_decay_steps = tc.max_iters + 2
    if it < tc.warmup_steps:
        return max_lr * (it + 1) / tc.warmup_steps
    if it > max_decay_steps:
        return min_lr
    ratio = min((it - tc.warmup_steps) / (max_decay_steps - tc.warmup_steps), 1.0)
    return min_lr + 0.5 * (1 + math.cos(math.pi * ratio)) * (max_lr - min_lr)


@torch.no_grad()
def estimate_loss(model, tc: Trainconfig, train_loader, val_loader):
    """Mean loss over ``eval_iters`` batches of each split (reference: single-gpu/train.py:280-293)."""
    out = {}
    model.eval()
    model.VAL_RUN = True
    for split, loader in (("train", train_loader), ("val", val_loader)):
        losses = torch.zeros(tc.eval_iters, device=loader.device)
        for k in range(tc.eval_iters):
            X, Y = loader.next_batch()
            _, loss, _ = model(X, Y)
            losses[k] = loss.detach().float()
        out[split] = losses.mean().item()
    model.train()
    model.VAL_RUN = False
    return out


def build_engine(mode: str, model: LLM, tc: Trainconfig, info: comm.DistInfo):
    kw = dict(weight_decay=tc.weight_decay, lr=tc.learning_rate, dtype=tc.dtype, bucket_mb=tc.bucket_mb,
              device=info.device, grad_dtype=tc.grad_dtype, vocab_parallel=getattr(tc, "vocab_parallel", False),
              tied_tail=getattr(tc, "tied_tail", "fused"), ep_capacity=getattr(tc, "ep_capacity", 0.0))
    if mode == "single":
        from .parallel.engine import Engine
        return Engine(model, **kw)
    if mode == "ddp":
        from .parallel.ddp import DDPEngine
        return DDPEngine(model, **kw)
    if mode in ("zero1", "zero2"):
        from .parallel.zero import ZeroEngine
        return ZeroEngine(model, stage=int(mode[-1]), **kw)
    if mode in ("fsdp", "fsdp_keep"):
        from .parallel.fsdp import FSDPEngine
        return FSDPEngine(model, reshard_after_forward=(mode == "fsdp"), prefetch=getattr(tc, "fsdp_prefetch", 1),
                          inflight=getattr(tc, "fsdp_inflight", 2), **kw)
    if mode == "ep":
        from .parallel.ep import EPEngine
        return EPEngine(model, **kw)
    if mode == "ep_fsdp":
        from .parallel.ep import EPFSDPEngine
        return EPFSDPEngine(model, prefetch=getattr(tc, "fsdp_prefetch", 1), inflight=getattr(tc, "fsdp_inflight", 2), **kw)
    raise ValueError(f"unknown parallel mode {mode!r}")


def train(tc: Trainconfig, mc: LLMconfig, mode: str = "single") -> dict:
    if mode != "single":
        comm.configure_rccl_env(tc.rccl_max_channels, tc.rccl_min_channels, tc.comm_priority,
                                backend=tc.comm_backend if torch.cuda.is_available() else None)
    info = comm.init_distributed() if mode != "single" else comm.DistInfo(
        0, 0, 1, torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu"), "none")
    master = info.is_master
    device = info.device
    torch.manual_seed(tc.seed + info.rank)
    if device.type == "cuda":
        torch.cuda.manual_seed(tc.seed + info.rank)

    def say(*a):
        if master:
            print(*a, flush=True)

    if mode != "single":
        say(f"DDP_WORLD_SIZE = {info.world_size}" if mode != "fsdp" else f"Num GPUs = {info.world_size}")

    # ---- data ------------------------------------------------------------------------------
    data_dir = resolve_data_dir(tc.dataset, tc.data_dir)
    if tc.synthetic_tokens > 0 or tc.dataset == "synthetic":
        n = tc.synthetic_tokens or max(4 * tc.total_batch_size, 1 << 20)
        data_dir = ensure_synthetic(data_dir, n, mc.vocab_size, info.rank, tc.seed)
    say(f"Using Dataset {os.path.basename(data_dir)}")
    B, T = tc.batch_size, mc.block_size
    for split in ("train.bin", "val.bin"):
        check_vocab(os.path.join(data_dir, split), mc.vocab_size)
    train_loader = DataLoader(B, T, os.path.join(data_dir, "train.bin"), device, info.rank, info.world_size, tc.seed)
    val_loader = DataLoader(B, T, os.path.join(data_dir, "val.bin"), device, info.rank, info.world_size, tc.seed + 7)

    # ---- gradient accumulation -------------------------------------------------------------
    denom = B * T * info.world_size
    assert tc.total_batch_size % denom == 0, "make sure total_batch_size is divisible by B * T * world_size"
    grad_accum_steps = tc.total_batch_size // denom

    # ---- model + engine --------------------------------------------------------------------
    model = LLM(mc)
    model.loss_grad_scale = 1.0 / grad_accum_steps     # every micro-batch loss is divided by this before backward
    total, active = model.get_num_params()
    say(f"total parameters = {total:,}, acitive parameters = {active:,}")
    flops_tok = model_flops_per_token(model, T)
    engine = build_engine(mode, model, tc, info)
    say("Using Fused AdamW")
    if mc.act_recomp:
        say("Using Activation Recomputation")
    optimizer = engine.configure_optimizers(weight_decay=tc.weight_decay, learning_rate=tc.learning_rate, device=device)

    start_iter = 0
    resume_path = tc.resume
    if resume_path == "auto":
        # restart-friendly: continue from this run's own periodic resume files if every rank has one
        # (torchrun --max-restarts relaunches the same command line after a failure), else start fresh
        cand = tc.file_name + "_resume"
        from .checkpoint import find_resume
        resume_path = cand if find_resume(cand, info.rank, device) is not False else ""
        say(f"--resume auto: {'found ' + cand if resume_path else 'no resume files, starting fresh'}")
        if not resume_path and master and os.path.exists(cand + ".latest"):
            # a marker without a complete generation: files deleted, or written by a job of another size whose ranks'
            # files are not all visible from here (node-local directories): say so instead of silently starting over
            say(f"warning: {cand}.latest exists but no generation it names is complete on every rank "
                f"(a restart on another number of ranks reads the files of ALL ranks of the job that wrote them)")
    if resume_path:
        start_iter = load_resume(resume_path, engine, (train_loader, val_loader), info.rank) + 1
        say(f"resumed from {resume_path} at iteration {start_iter}")

    # Launch-bound steps (few tokens per step) are replayed from one hipGraph, the role torch.compile
    # plays in the reference trainer (reference: single-gpu/train.py:262-266). "auto" turns it on for
    # one GPU and at most 32K tokens per step; "on" also captures the data-parallel engine (all-reduces
    # included); a failed capture falls back to eager.
    graphed = None
    mode_g = str(tc.hip_graph).lower()
    want_graph = mode_g in ("on", "true", "1") or (
        mode_g == "auto" and tc.total_batch_size <= 32768 and not tc.profile)
    on = mode_g in ("on", "true", "1")
    capturable = (not engine._dist and type(engine).name == "single") or (on and type(engine).name == "ddp")
    if want_graph and device.type == "cuda" and capturable:
        from .parallel.graph import GraphedTrainStep
        graphed = GraphedTrainStep(engine, grad_accum_steps, tc.grad_clip)
        say("Using hipGraph capture of the optimizer step")

    sink = JsonlSink(tc.metrics_jsonl if master else "")
    timer = PhaseTimer(tc.profile)
    from .utils import profiling
    profiling.set_active(timer if timer.enabled else None)
    ktable = profiling.KernelTable(timer.mode == "kernels" and device.type == "cuda")
    train_loss_stats, valrun_val, valrun_train = [], [], []
    x, y = train_loader.next_batch()
    last_loss = float("nan")
    n_skipped = 0
    tokens_per_s = 0.0
    # Fault injection for restart drills: DPA_FAULT_AT_ITER=<n> makes the last rank die once at iteration n
    # (a marker file keeps the relaunched job from dying again).
    fault_at = int(os.environ.get("DPA_FAULT_AT_ITER", "-1"))
    win_t0, win_n, dt = None, 0, float("nan")     # timing window since the last host <- device synchronisation
    for it in range(start_iter, tc.max_iters + 1):
        if it == fault_at and info.rank == info.world_size - 1:
            marker = tc.file_name + "_fault_injected"
            if not os.path.exists(marker):
                open(marker, "w").close()
                print(f"[fault injection] rank {info.rank} exits at iteration {it}", flush=True)
                os._exit(17)
        t0 = perf_counter()
        lr = get_lr(it, tc)
        for pg in optimizer.param_groups:
            pg["lr"] = lr
        optimizer.zero_grad(set_to_none=True)

        if tc.eval and (it % tc.eval_interval == 0 or it == tc.max_iters) and it != 0:
            a = perf_counter()
            losses = estimate_loss(engine, tc, train_loader, val_loader)
            valrun_val.append(losses["val"])
            valrun_train.append(losses["train"])
            b = perf_counter()
            say(f"--------val run-------- train loss {losses['train']:.4f} | val loss {losses['val']:.4f} | "
                f"dt {1000 * (b - a):.4f}ms")
            t0 = b
            win_t0, win_n = None, 0     # (the evaluation drained the device: the timing window starts over)

        if win_t0 is None:
            win_t0 = t0
        ktable.step_begin()
        if graphed is not None:
            batches = [(x, y)]
            for _ in range(grad_accum_steps - 1):
                batches.append(train_loader.next_batch())
            loader_state = train_loader.state_dict()
            with timer.phase("graph"):
                loss = graphed.step(batches, lr) / grad_accum_steps
            x, y = train_loader.next_batch()
        for micro_step in range(grad_accum_steps if graphed is None else 0):
            engine.require_backward_grad_sync = (micro_step == grad_accum_steps - 1)
            with timer.phase("forward"):
                _, loss, _ = engine(x, y)
                loss = loss / grad_accum_steps
            loader_state = train_loader.state_dict()   # state from which the prefetched batch is redrawn
            x, y = train_loader.next_batch()   # host work overlaps the queued GPU work
            with timer.phase("backward"):
                loss.backward()

        if graphed is None:
            with timer.phase("optimizer"):
                if tc.grad_clip != 0.0:
                    engine.clip_grad_norm_(tc.grad_clip)
                optimizer.step()

        ktable.step_end()
        read_loss = (it % max(tc.sync_loss_every, 1) == 0) or it == tc.max_iters
        if read_loss:
            last_loss = loss.item() * grad_accum_steps   # the only host<-device sync of the step
            train_loss_stats.append(last_loss)
            comm.check_comm_health()
            skipped = engine.skipped_steps()
            if skipped > n_skipped:
                say(f"warning: {skipped - n_skipped} optimizer step(s) skipped for a non-finite gradient norm "
                    f"({skipped} so far)")
                n_skipped = skipped
        mem = 0
        if device.type == "cuda":
            if read_loss:
                torch.cuda.synchronize()
            mem = torch.cuda.memory_reserved()
        # Step time: wall clock since the last point at which the device was drained, per step of that window. With
        # --sync_loss_every 1 (the reference's behaviour) that is this step's own wall time; with N > 1 the steps in
        # between only enqueue work (microseconds of wall time) and the N-th one waits for all of them, so neither is
        # a step time by itself: the window's average is, and the steps in between repeat the last measured value.
        win_n += 1
        if read_loss or device.type != "cuda":
            dt = (perf_counter() - win_t0) / win_n
            win_t0, win_n = None, 0
        tokens_per_s = tc.total_batch_size / dt
        util = mfu(tokens_per_s, flops_tok, info.world_size)
        if it % max(tc.log_interval, 1) == 0:
            say(f"step: {it} | train loss:{last_loss:.4f} | dt: {1000 * dt:.2f}ms | "
                f"grad_accum_steps: {grad_accum_steps} | GPU RAM: {mem / 1024 ** 3:.2f}GB | "
                f"tok/s: {tokens_per_s:,.0f} | MFU: {100 * util:.2f}%")
        sink.write(step=it, loss=last_loss, dt_ms=1000 * dt, tokens_per_s=tokens_per_s, mfu=util, lr=lr,
                   mem_gb=mem / 1024 ** 3, peak_flops=PEAK_BF16_DENSE_FLOPS)
        poll_pending_save(device)   # (--save_async: commits the generation in flight once every rank has written it)
        if tc.save_interval and it > 0 and it % tc.save_interval == 0:
            save_resume(tc.file_name + "_resume", engine, it, (train_loader, val_loader), info.rank, info.world_size,
                        loader_states=[loader_state, val_loader.state_dict()], background=getattr(tc, "save_async", False))

    finish_pending_save()       # (--save_async: the last generation's write and commit)
    if tc.profile and master:
        print(timer.report(), flush=True)
        if ktable.table:
            print(ktable.table, flush=True)
    profiling.set_active(None)
    result = {"final_loss": last_loss, "tokens_per_s": tokens_per_s, "losses": train_loss_stats,
              "grad_accum_steps": grad_accum_steps, "total_params": total, "active_params": active}
    if tc.save_model:
        loss_stats = {"train": train_loss_stats, "valrun_val": valrun_val, "valrun_train": valrun_train}
        path = save_reference_checkpoint(engine, mc, tc, tc.file_name, loss_stats, total, active, info.rank)
        if master:
            say("Model and config saved to {}.pt".format(tc.file_name + "_ckpt"))
            say("Stats and config saved to {}.pt".format(tc.file_name + "_stats"))
            result["checkpoint"] = path
    sink.close()
    if mode != "single":
        comm.barrier()
        comm.shutdown()
    return result
