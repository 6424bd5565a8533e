"""Phase markers and timing for ``--profile`` runs.

``PhaseTimer.phase(name)`` opens a roctx range (visible in ``rocprofv3 --marker-trace`` timelines) when
the roctx library shipped with PyTorch-ROCm is loadable, and accounts the phase's time in one of
two ways:

* ``events`` (default of ``--profile``): a HIP event is recorded on the current stream at both ends of
  the phase and the elapsed times are read once, at report time. Nothing synchronises inside the
  training loop, so the overlap of collectives with backward that the run is meant to show survives;
  a phase's time is the GPU time between its two events on the compute stream.
* ``sync``: the device is synchronised at both ends and wall-clock is accumulated. Exact per-phase
  wall time, but it serialises phases that normally overlap.

Communication is reported next to the compute phases (forward / backward / optimizer):
``comm_issue`` ranges wrap every collective call of ``parallel/comm.py`` (host time to enqueue, with
byte counts), and ``comm_wait`` is the time the compute stream spends waiting for outstanding
collectives at the end of backward, i.e. the communication that was NOT hidden.

``KernelTable`` (``--profile kernels``) collects per-kernel device time over a few steps through
``torch.profiler`` (roctracer underneath) and prints the same per-kernel ms/step table that
``benchmarks/summarize_prof.py`` derives from a ``rocprofv3 --kernel-trace --stats`` run.

The reference has no profiling hooks at all (it times whole steps with ``perf_counter``,
reference: single-gpu/train.py:321,358).
"""
from __future__ import annotations

import contextlib
import ctypes
import time
from collections import defaultdict

import torch

_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is None:
        try:
            _roctx = ctypes.CDLL("libroctx64.so")
        except OSError:
            _roctx = False
    return _roctx


def normalize_mode(profile) -> str:
    """'' (off) | 'events' | 'sync' | 'kernels' from a flag value (bools of older configs included)."""
    if profile is True:
        return "events"
    if not profile:
        return ""
    mode = str(profile).lower()
    if mode in ("1", "true", "on"):
        return "events"
    assert mode in ("events", "sync", "kernels"), f"--profile {profile!r}: events | sync | kernels"
    return mode


class PhaseTimer:
    def __init__(self, enabled):
        self.mode = normalize_mode(enabled)
        self.enabled = self.mode in ("events", "sync")
        self.tot = defaultdict(float)      # seconds
        self.cnt = defaultdict(int)
        self.host = defaultdict(float)     # host-side seconds of phases that only enqueue (comm_issue)
        self.bytes = defaultdict(int)
        self._pending = []                 # (name, start event, end event)

    @contextlib.contextmanager
    def phase(self, name: str, nbytes: int = 0, host_only: bool = False):
        if not self.enabled:
            yield
            return
        lib = _load_roctx()
        cuda = torch.cuda.is_available()
        if lib:
            lib.roctxRangePushA(name.encode())
        e0 = e1 = None
        if self.mode == "sync" and cuda and not host_only:
            torch.cuda.synchronize()
        elif self.mode == "events" and cuda and not host_only:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if host_only:
                self.host[name] += time.perf_counter() - t0
            elif e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._pending.append((name, e0, e1))
                if len(self._pending) >= 256:
                    self._drain_finished()
            else:
                if cuda:
                    torch.cuda.synchronize()
                self.tot[name] += time.perf_counter() - t0
            self.cnt[name] += 1
            self.bytes[name] += nbytes
            if lib:
                lib.roctxRangePop()

    def _drain_finished(self):
        """Fold the event pairs that have completed into the totals without waiting for the others: a long profiled run
        then holds a few hundred events at any time instead of twenty more per step."""
        keep = []
        for name, e0, e1 in self._pending:
            if e1.query():
                self.tot[name] += e0.elapsed_time(e1) * 1e-3
            else:
                keep.append((name, e0, e1))
        self._pending = keep

    def _resolve(self):
        if self._pending:
            torch.cuda.synchronize()
            for name, e0, e1 in self._pending:
                self.tot[name] += e0.elapsed_time(e1) * 1e-3
            self._pending = []

    def report(self) -> str:
        self._resolve()
        if not self.tot and not self.host:
            return ""
        total = sum(self.tot.values()) or 1.0
        how = "GPU time between stream events, no synchronisation" if self.mode == "events" else "synchronised wall clock"
        lines = [f"phase ({how})      calls    total ms    share"]
        for k, v in sorted(self.tot.items(), key=lambda kv: -kv[1]):
            lines.append(f"{k:<24} {self.cnt[k]:>6} {1000 * v:>11.2f} {100 * v / total:>7.1f}%")
        for k, v in sorted(self.host.items(), key=lambda kv: -kv[1]):
            mb = self.bytes[k] / 2 ** 20
            lines.append(f"{k:<24} {self.cnt[k]:>6} {1000 * v:>11.2f}   (host enqueue time, {mb:,.1f} MiB moved)")
        lines += self._comm_busy_lines()
        return "\n".join(lines)

    @staticmethod
    def _comm_busy_lines():
        """Device time of each collective on the communication stream (own RCCL communicator only):
        together with ``comm_wait`` it tells how much of the communication ran hidden under compute."""
        from ..parallel import comm
        busy, nbytes, calls = {}, {}, {}
        for kind, b, ms in comm.drain_comm_timings():
            busy[kind] = busy.get(kind, 0.0) + ms
            nbytes[kind] = nbytes.get(kind, 0) + b
            calls[kind] = calls.get(kind, 0) + 1
        out = []
        for k, ms in sorted(busy.items(), key=lambda kv: -kv[1]):
            gbs = nbytes[k] / 1e9 / (ms * 1e-3) if ms > 0 else 0.0
            out.append(f"{'comm_busy:' + k:<24} {calls[k]:>6} {ms:>11.2f}   (on the communication stream, "
                       f"{nbytes[k] / 2 ** 20:,.1f} MiB, {gbs:,.1f} GB/s)")
        return out


# The timer the communication layer reports to (set by the trainer for the duration of a --profile run).
_active: PhaseTimer | None = None


def set_active(timer: PhaseTimer | None):
    global _active
    _active = timer
    from ..parallel import comm
    comm.set_comm_timing(timer is not None and timer.enabled)


def comm_range(kind: str, nbytes: int = 0):
    """Context of one collective call: roctx range ``comm:<kind>`` + host enqueue time + bytes."""
    if _active is None or not _active.enabled:
        return contextlib.nullcontext()
    return _active.phase(f"comm_issue:{kind}", nbytes=nbytes, host_only=True)


class ExposedComm:
    """Exposed communication without a profiling run: while ``start()`` is in effect, every wait of the compute
    stream for outstanding collectives (the ``comm_wait_range`` sections of the engines) is bracketed by two HIP
    events on that stream; ``stop()`` reads them once and returns the total milliseconds the stream spent there.
    No host synchronisation happens inside the measured region (bench.py reports it per step)."""

    def __init__(self):
        self.on = False
        self.pairs = []

    def start(self):
        self.pairs = []
        self.on = torch.cuda.is_available()

    @contextlib.contextmanager
    def section(self):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            yield
        finally:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.append((e0, e1))

    def stop(self) -> float:
        self.on = False
        if not self.pairs:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in self.pairs)
        self.pairs = []
        return ms


exposed_comm = ExposedComm()


def comm_wait_range():
    """Context of the compute stream waiting for outstanding collectives (exposed communication)."""
    if _active is not None and _active.enabled:
        return _active.phase("comm_wait")
    if exposed_comm.on:
        return exposed_comm.section()
    return contextlib.nullcontext()


class KernelTable:
    """Per-kernel device time over the profiled steps (``--profile kernels``)."""

    def __init__(self, enabled: bool, skip: int = 3, steps: int = 4):
        self.enabled = enabled
        self.skip, self.steps = skip, steps
        self._seen = 0
        self._prof = None
        self.table = ""

    def step_begin(self):
        if not self.enabled or self.table:
            return
        if self._seen == self.skip:
            from torch.profiler import ProfilerActivity, profile
            self._prof = profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU])
            self._prof.__enter__()
        self._seen += 1

    def step_end(self):
        if self._prof is None or self._seen < self.skip + self.steps:
            return
        torch.cuda.synchronize()
        self._prof.__exit__(None, None, None)
        rows = []
        for ev in self._prof.key_averages():
            dev_us = getattr(ev, "device_time_total", None)
            if dev_us is None:
                dev_us = getattr(ev, "cuda_time_total", 0.0)
            if getattr(ev, "device_type", None) is not None and "CUDA" in str(ev.device_type) and dev_us > 0:
                rows.append((dev_us, ev.count, ev.key))
        self._prof = None
        rows.sort(reverse=True)
        tot = sum(r[0] for r in rows) or 1.0
        lines = [f"total GPU kernel time per step: {tot / 1e3 / self.steps:.2f} ms  ({self.steps} steps profiled)"]
        for us, cnt, name in rows[:40]:
            lines.append(f"{us / 1e3 / self.steps:8.3f} ms/step {100 * us / tot:6.2f}%  calls/step {cnt / self.steps:7.1f}  "
                         f"avg {us / max(cnt, 1):9.1f} us  {name[:100]}")
        self.table = "\n".join(lines)
