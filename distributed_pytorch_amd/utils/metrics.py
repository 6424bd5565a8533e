"""Throughput accounting: FLOPs per token, MFU, JSONL metric sink.

``FLOPs/token = 6 N + 12 L C T`` with N = parameters excluding the learned position table and
the tied embedding counted once; the attention term is not halved for causality (nanoGPT /
PaLM convention). MFU is quoted against the MI355X *dense* bf16 matrix peak below — the
vendor's larger figures include 2:1 structured sparsity and are never used here.
"""
from __future__ import annotations

import json
import time

PEAK_BF16_DENSE_FLOPS = 2.5e15   # MI355X dense bf16 MFMA peak per GPU (datasheet class figure)


def flops_per_token(n_params_total: int, n_layer: int, n_embd: int, seq_len: int, pos_table_params: int = 0,
                    n_params_active: int | None = None) -> float:
    n = (n_params_active if n_params_active is not None else n_params_total) - pos_table_params
    return 6.0 * n + 12.0 * n_layer * n_embd * seq_len


def model_flops_per_token(model, seq_len: int) -> float:
    cfg = model.config
    total, active = model.get_num_params()
    pos = cfg.block_size * cfg.n_embd if cfg.pos_emb == "learn" else 0
    return flops_per_token(total, cfg.n_layer, cfg.n_embd, seq_len, pos, active)


def mfu(tokens_per_s: float, flops_tok: float, n_gpus: int, peak: float = PEAK_BF16_DENSE_FLOPS) -> float:
    return tokens_per_s * flops_tok / (n_gpus * peak)


class JsonlSink:
    def __init__(self, path: str):
        self.f = open(path, "a") if path else None

    def write(self, **kv):
        if self.f:
            kv.setdefault("time", time.time())
            self.f.write(json.dumps(kv) + "\n")
            self.f.flush()

    def close(self):
        if self.f:
            self.f.close()
