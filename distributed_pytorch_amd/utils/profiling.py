"""Lightweight phase markers for ``--profile`` runs.

``phase(name)`` opens a roctx range (visible in ``rocprofv3 --marker-trace`` timelines) when
the roctx library shipped with PyTorch-ROCm is loadable, and always accumulates wall-clock
per phase (after a device synchronize) into ``PhaseTimer`` for the end-of-run report.
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


class PhaseTimer:
    def __init__(self, enabled: bool):
        self.enabled = enabled
        self.tot = defaultdict(float)
        self.cnt = defaultdict(int)

    @contextlib.contextmanager
    def phase(self, name: str):
        if not self.enabled:
            yield
            return
        lib = _load_roctx()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if lib:
            lib.roctxRangePushA(name.encode())
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.tot[name] += time.perf_counter() - t0
            self.cnt[name] += 1
            if lib:
                lib.roctxRangePop()

    def report(self) -> str:
        if not self.tot:
            return ""
        total = sum(self.tot.values())
        lines = ["phase                 calls    total ms    share"]
        for k, v in sorted(self.tot.items(), key=lambda kv: -kv[1]):
            lines.append(f"{k:<20} {self.cnt[k]:>6} {1000 * v:>11.2f} {100 * v / total:>7.1f}%")
        return "\n".join(lines)
