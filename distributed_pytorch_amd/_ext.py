"""Loader of the in-tree gfx950 extension (``_C.so``, built by ``build.py``).

``ops()`` returns ``torch.ops.dpa``. On a machine with a GPU a missing extension is an error —
the framework never falls back silently to eager PyTorch kernels there. On a CPU-only machine
the library is optional: the eager reference implementations in ``ops/`` carry the plumbing
tests.
"""
from __future__ import annotations

import os
from pathlib import Path

import torch

# DPA_EXT_LIB names another build of the library (A/B timing of two builds on one machine).
_LIB = Path(os.environ.get("DPA_EXT_LIB") or Path(__file__).resolve().parent / "_C.so")
_loaded = False
_error: Exception | None = None


def load() -> bool:
    global _loaded, _error
    if _loaded:
        return True
    if _error is not None:
        return False
    if not _LIB.exists():
        _error = FileNotFoundError(f"{_LIB} not found: run `python build.py`")
        return False
    try:
        torch.ops.load_library(str(_LIB))
        _loaded = True
    except Exception as e:  # pragma: no cover - depends on the machine
        _error = e
    return _loaded


def available() -> bool:
    return load()


def require():
    if not load():
        raise RuntimeError(
            "the gfx950 HIP extension is required for GPU tensors but could not be loaded "
            f"({_error!r}); build it with `python build.py`")
    return torch.ops.dpa


def ops():
    return require()


def use_hip(*tensors: torch.Tensor) -> bool:
    """True when the HIP kernels must handle these tensors (all on the GPU). Raises if they are
    on the GPU but the extension is missing."""
    if not tensors or not all(t.is_cuda for t in tensors if t is not None):
        return False
    if os.environ.get("DPA_FORCE_EAGER", "0") == "1":
        return False
    require()
    return True
