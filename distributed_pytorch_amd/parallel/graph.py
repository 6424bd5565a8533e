"""A whole optimizer step captured in one hipGraph.

Small models are bound by kernel-launch latency, not by the GPU: a 6-layer network issues a few
hundred kernels of a few microseconds each per step. Capturing the step (zero the flat gradient
buffers -> ``accum`` x (forward + backward) -> global-norm clip -> fused AdamW) once and
replaying it removes the per-kernel launch cost from every later step. This is the role
``torch.compile`` plays in the reference trainers (reference: single-gpu/train.py:262-266,
multi-gpu/ddp/train.py:276-280) expressed with HIP graphs instead of a tracing compiler: the
kernels are the same hand-written ones, only their submission is batched.

What makes the step capturable:
  * every shape is static and nothing in it reads a value back to the host (the clip threshold
    test and the non-finite-gradient skip happen inside the AdamW kernel);
  * the step-dependent scalars (learning rate, Adam bias corrections) live in a 3-float device
    tensor that is refreshed before each replay, because kernel arguments are frozen at capture;
  * gradients accumulate into the engine's flat buffers, whose addresses never change.

The plain engine and the data-parallel engine can be captured. For DDP the bucketed all-reduces are
part of the graph (RCCL collectives are capturable; they run on the communicator's stream, which the
capture follows through the events that order it with the compute stream), so a launch-bound model
keeps its advantage on several GPUs. The sharded engines (ZeRO, FSDP, EP) are not captured: their
storage is resized and their collectives depend on host-side bookkeeping inside a step.
"""
from __future__ import annotations

import math

import torch


def _clear_runtime_error() -> None:
    """An invalidated capture leaves the runtime's sticky last-error set; the next launch's error check would
    report it as its own failure. Read it away (and drop what the failed capture left in the allocator)."""
    try:
        from .._ext import ops as _ops
        _ops().clear_last_error()
    except Exception:
        pass
    try:
        torch.cuda.synchronize()
    except Exception:
        try:
            from .._ext import ops as _ops
            _ops().clear_last_error()
        except Exception:
            pass


class GraphedTrainStep:
    """``step(batches, lr) -> loss`` where ``batches`` is a list of ``accum`` ``(x, y)`` pairs of shape
    ``[B, T]``. The first ``warmup`` calls run eagerly (they are real training steps and let every
    lazy initialisation happen), the next call captures, all later calls replay."""

    def __init__(self, engine, accum: int = 1, grad_clip: float = 1.0, warmup: int = 3):
        assert engine.device.type == "cuda", "hipGraph capture needs the GPU"
        assert (not engine._dist) or type(engine).name == "ddp", \
            "only the plain and the data-parallel engine are captured"
        self.engine = engine
        self.accum = int(accum)
        self.grad_clip = float(grad_clip)
        self.warmup = int(warmup)
        self.calls = 0
        self.failed = False        # capture raised once: stay eager
        self.graph = None
        self.x = self.y = None
        self.loss = None
        self._dyn = torch.zeros(3, dtype=torch.float32, device=engine.device)

    # -- pieces -----------------------------------------------------------------------------
    def _set_scalars(self, lr: float):
        b1, b2 = self.engine.betas
        t = self.engine.step_count + 1
        # fill kernels carry the values as launch arguments: no host buffer that a later step could
        # overwrite while an earlier asynchronous copy is still queued
        self._dyn[0].fill_(lr)
        self._dyn[1].fill_(1.0 - b1 ** t)
        self._dyn[2].fill_(math.sqrt(1.0 - b2 ** t))

    def _body(self, xs, ys):
        eng = self.engine
        eng.zero_grad()
        loss = None
        for i in range(self.accum):
            eng.require_backward_grad_sync = i == self.accum - 1     # DDP: reduce with the last micro-batch
            _, loss, _ = eng(xs[i], ys[i])
            loss = loss / self.accum
            loss.backward()
        if self.grad_clip != 0.0:
            eng.clip_grad_norm_(self.grad_clip)
        eng.step(0.0)     # lr and bias corrections come from the device scalars
        return loss.detach()

    # -- public -----------------------------------------------------------------------------
    def step(self, batches, lr: float) -> torch.Tensor:
        assert len(batches) == self.accum
        eng = self.engine
        eng._dyn = self._dyn
        try:
            if self.failed or (self.graph is None and self.calls < self.warmup):
                self._set_scalars(lr)
                loss = self._body([b[0] for b in batches], [b[1] for b in batches])
                self.calls += 1
                return loss * self.accum   # the last micro-batch's own loss
            if self.graph is None:
                self.x = torch.stack([b[0] for b in batches]).clone()
                self.y = torch.stack([b[1] for b in batches]).clone()
                count = eng.step_count
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                try:
                    # thread_local: other threads of the process (the collective watchdog polling its
                    # events, the pinned-memory loader) keep calling the runtime while this one captures;
                    # under the default global mode such a call aborts the capture, or the process
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        self.loss = self._body(list(self.x.unbind(0)), list(self.y.unbind(0)))
                except Exception as exc:      # something in this configuration cannot be captured
                    eng.step_count = count
                    self.failed = True
                    self.x = self.y = self.loss = None
                    _clear_runtime_error()
                    torch.cuda.synchronize()
                    print(f"[hip_graph] capture failed ({type(exc).__name__}: {exc}); continuing eagerly", flush=True)
                    self._set_scalars(lr)
                    loss = self._body([b[0] for b in batches], [b[1] for b in batches])
                    self.calls += 1
                    return loss * self.accum
                self.graph = graph
                eng.step_count = count      # capture records work, it does not run it
            else:
                for i, (bx, by) in enumerate(batches):
                    self.x[i].copy_(bx, non_blocking=True)
                    self.y[i].copy_(by, non_blocking=True)
            self._set_scalars(lr)
            self.graph.replay()
            eng.step_count += 1
            self.calls += 1
            return self.loss * self.accum
        finally:
            eng._dyn = None
