"""Checkpoints.

``save_reference_checkpoint`` writes the reference's two files
(reference: single-gpu/train.py:361-372):

    <name>_ckpt.pt  = {'model_config', 'train_config', 'model_state'}
    <name>_stats.pt = {'model_config', 'train_config', 'losses', 'total_params', 'active_params'}

``model_state`` uses the un-prefixed key schema of the reference ``LLM`` and fp32 tensors, so
``reference.LLM(cfg).load_state_dict(ckpt['model_state'], strict=True)`` works; the configs are
stored both as dataclass instances (like the reference) and as plain dicts
(``*_config_dict``) that load with ``weights_only=True``.

``save_resume`` / ``load_resume`` add what the reference lacks: optimizer state (per rank for
sharded engines), iteration counter, RNG and data-loader state.
"""
from __future__ import annotations

import os
from pathlib import Path

import torch

from .parallel import comm


def save_reference_checkpoint(engine_or_model, model_config, train_config, file_name: str, losses=None,
                              total_params=None, active_params=None, rank: int = 0):
    state = engine_or_model.state_dict()          # collective for sharded engines: call on every rank
    if rank != 0:
        return None
    state = {k: v.detach().to("cpu") for k, v in state.items()}
    ckpt = {"model_config": model_config, "train_config": train_config, "model_state": state,
            "model_config_dict": model_config.to_dict(), "train_config_dict": train_config.to_dict()}
    torch.save(ckpt, file_name + "_ckpt.pt")
    stats = {"model_config": model_config, "train_config": train_config, "losses": losses or {},
             "total_params": total_params, "active_params": active_params,
             "model_config_dict": model_config.to_dict(), "train_config_dict": train_config.to_dict()}
    torch.save(stats, file_name + "_stats.pt")
    return file_name + "_ckpt.pt"


def save_resume(path: str, engine, it: int, loaders=(), rank: int = 0, world: int = 1, extra=None,
                loader_states=None):
    """Per-rank resume file ``<path>.rank<r>`` (optimizer shards are rank-local)."""
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    payload = {
        "iter": it,
        "world": world,
        "optimizer": engine.optimizer_state_dict(),
        "buffers": {k: v.detach().cpu() for k, v in engine.module.named_buffers() if v is not None},
        "torch_rng": torch.get_rng_state(),
        "cuda_rng": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
        "loaders": loader_states if loader_states is not None else [l.state_dict() for l in loaders],
        "extra": extra or {},
    }
    tmp = f"{path}.rank{rank}.tmp"
    torch.save(payload, tmp)
    os.replace(tmp, f"{path}.rank{rank}")
    comm.barrier()


def load_resume(path: str, engine, loaders=(), rank: int = 0) -> int:
    payload = torch.load(f"{path}.rank{rank}", map_location=engine.device, weights_only=False)
    engine.load_optimizer_state_dict(payload["optimizer"])
    with torch.no_grad():
        bufs = dict(engine.module.named_buffers())
        for k, v in payload["buffers"].items():
            if k in bufs and bufs[k] is not None:
                bufs[k].copy_(v.to(bufs[k].device))
    torch.set_rng_state(payload["torch_rng"].cpu())
    if payload["cuda_rng"] is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(payload["cuda_rng"].cpu())
    for l, sd in zip(loaders, payload["loaders"]):
        l.load_state_dict(sd)
    return int(payload["iter"])
