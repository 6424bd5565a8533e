"""Next-token sampling: temperature, top-k filter, softmax and one multinomial draw.

GPU: one HIP kernel per call (``csrc/sample.hip``), seeded by one device integer drawn from torch's
generator of that device, so it is reproducible under ``manual_seed`` and can be captured in a hipGraph
(every replay draws a new seed). CPU / fallback: the reference's chain of stock ops
(reference: single-gpu/model.py:733-745)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip


def sample_eager(logits: torch.Tensor, temperature: float = 1.0, topk: int | None = None) -> torch.Tensor:
    """``logits [B, V]`` -> token ids ``[B, 1]`` with stock ops (no host-built index tensors)."""
    logits = logits.float()
    if temperature > 0:
        logits = logits / temperature
    if topk is not None:
        v, _ = torch.topk(logits, min(topk, logits.size(-1)))
        logits = logits.masked_fill(logits < v[:, -1:], -float("inf"))
    return torch.multinomial(F.softmax(logits, dim=-1), num_samples=1)


def _fused_ok(logits: torch.Tensor) -> bool:
    V = logits.shape[-1]
    return (use_hip(logits) and logits.dim() == 2 and logits.dtype in (torch.bfloat16, torch.float32)
            and logits.stride(1) == 1 and V % 8 == 0 and 8 <= V <= 57344
            and (logits.stride(0) * logits.element_size()) % 16 == 0 and logits.data_ptr() % 16 == 0)


def sample_step_supported(logits: torch.Tensor) -> bool:
    return _fused_ok(logits) and hasattr(ops(), "sample_topk_step")


def sample_step(logits: torch.Tensor, temperature: float, topk: int | None, seed: torch.Tensor, tok: torch.Tensor,
                history: torch.Tensor, history_pos: torch.Tensor, pos: torch.Tensor, len_after: torch.Tensor,
                ticket: torch.Tensor) -> None:
    """Sampling and the bookkeeping of one captured decode step in one launch: the drawn tokens go to ``tok
    [B, 1]`` (the next step's input) and to ``history[:, history_pos]``; the device counters ``pos``,
    ``len_after`` and ``history_pos`` then advance by one. ``seed`` is fixed for the generation (the kernel
    mixes ``pos`` into the random stream), ``ticket`` a zeroed int32 ``[1]`` the kernel keeps at zero."""
    ops().sample_topk_step(logits, float(temperature), int(topk or 0), seed, tok, history, history_pos, pos, len_after,
                           ticket)


def sample_fused(logits: torch.Tensor, temperature: float = 1.0, topk: int | None = None,
                 seed: torch.Tensor | None = None) -> torch.Tensor:
    """Same distribution from one kernel launch. ``seed``: int64 ``[1]`` on the device (default: drawn)."""
    if not _fused_ok(logits):
        return sample_eager(logits, temperature, topk)
    if seed is None:
        seed = torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64, device=logits.device)
    return ops().sample_topk(logits, float(temperature), int(topk or 0), seed)
