"""Token data: memory-mapped ``uint16`` streams sampled by random crops.

``DataLoader.next_batch()`` has the reference's semantics — ``B`` independent random windows of
``T + 1`` tokens from a flat token file, ``x = w[:-1]``, ``y = w[1:]``, int64, moved to the
device through pinned memory (reference: single-gpu/train.py:210-254). Two deliberate changes:

* sampling uses a private generator and is *globally* deterministic: every call draws the
  start offsets of the whole global micro-batch (``B * world`` windows) from a generator
  seeded identically on all ranks, and each rank keeps its own slice. A ``W``-rank run with
  micro-batch ``B`` therefore consumes exactly the windows of a 1-rank run with micro-batch
  ``W * B`` (the reference decorrelates ranks only through ``manual_seed(1729 + rank)``);
* the next batch is assembled into a reusable pinned staging buffer and copied on a dedicated
  copy stream, so the host work and the transfer overlap the GPU work already queued.

``ensure_synthetic`` writes a synthetic token stream of the Shakespeare / TinyStories shape
(GPT-2-BPE-like ids < vocab_size with a Zipf-like unigram distribution, EOT every few hundred
tokens) for machines without the datasets. Rank 0 writes, everybody else waits on a barrier.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

from .parallel import comm


class DataLoader:
    def __init__(self, B: int, T: int, file_path: str, device, rank: int = 0, world: int = 1, seed: int = 1729):
        self.B, self.T = B, T
        self.file_path = str(file_path)
        self.device = torch.device(device)
        self.rank, self.world = rank, world
        self.tokens = np.memmap(self.file_path, dtype=np.uint16, mode="r")
        self.N = len(self.tokens)
        self._native = None
        try:   # native mmap reader + threaded batch assembly (csrc/dataloader.cpp) when the extension is built
            from . import _ext
            if _ext.load():
                self._native = torch.classes.dpa.TokenFile(self.file_path)
        except Exception:
            self._native = None
        if self.B * self.T + 1 > self.N:
            raise ValueError(f"Batch size {B} and block size {T} are too large for dataset of length {self.N}")
        self.gen = torch.Generator()
        self.gen.manual_seed(seed)
        self._pinned = None
        if self.device.type == "cuda":
            # two pinned staging buffers and a copy stream of its own: the host-to-device transfer of the
            # next batch runs beside the compute stream's kernels instead of queueing behind them
            self._pinned = [torch.empty(2, B, T, dtype=torch.long).pin_memory() for _ in range(2)]
            self._copied = [None, None]       # event of the last transfer out of each staging buffer
            self._flip = 0
            self._copy_stream = torch.cuda.Stream(device=self.device)

    def state_dict(self):
        return {"gen": self.gen.get_state()}

    def load_state_dict(self, sd):
        self.gen.set_state(sd["gen"].cpu())   # a resume file may have been mapped to the GPU

    def _starts(self) -> np.ndarray:
        s = torch.randint(0, self.N - self.T - 1, (self.B * self.world,), generator=self.gen)
        return s[self.rank * self.B:(self.rank + 1) * self.B].numpy()

    def next_batch(self):
        B, T = self.B, self.T
        starts = self._starts()
        slot = None
        if self._pinned is not None:
            slot = self._flip
            buf = self._pinned[slot]
            self._flip ^= 1
            if self._copied[slot] is not None:
                self._copied[slot].synchronize()   # the transfer that last read this buffer has finished
        else:
            buf = torch.empty(2, B, T, dtype=torch.long)
        if self._native is not None:
            self._native.gather(torch.from_numpy(np.ascontiguousarray(starts)), T, buf, 1 if B * T <= (1 << 20) else 4)
        else:
            arr = buf.numpy()
            for i, s in enumerate(starts):
                w = self.tokens[s:s + T + 1]
                arr[0, i] = w[:-1]
                arr[1, i] = w[1:]
        if self.device.type == "cuda":
            with torch.cuda.stream(self._copy_stream):
                dev = buf.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
            self._copied[slot] = done
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)            # consumers on the compute stream are ordered after the copy
            dev.record_stream(cur)          # the allocator must not recycle it while compute still reads it
            return dev[0], dev[1]
        return buf[0], buf[1]


def synthetic_tokens(n: int, vocab_size: int, seed: int = 1729) -> np.ndarray:
    """A deterministic uint16 token stream with a Zipf-like unigram distribution and an
    end-of-text token (50256 when the vocabulary has it) roughly every 250 tokens. ``seed`` picks the
    sample; which token has which rank is the same for every seed, so that a validation split drawn
    with another seed comes from the distribution the training split was drawn from."""
    rng = np.random.default_rng(seed)
    v = min(vocab_size, 65535)
    ranks = np.arange(1, v + 1, dtype=np.float64)
    p = 1.0 / ranks ** 1.05
    p /= p.sum()
    rng.permutation(v)                                   # (keeps the sample stream of seed 1729 as it always was)
    perm = np.random.default_rng(1729).permutation(v)
    toks = perm[rng.choice(v, size=n, p=p)].astype(np.uint16)
    eot = 50256 if vocab_size > 50256 else v - 1
    toks[rng.random(n) < 1.0 / 250] = eot
    return toks


def check_vocab(file_path: str, vocab_size: int, probe_tokens: int = 1 << 22):
    """Fail early, on the host, when a token file holds ids the model has no row for (a file tokenised
    for another vocabulary). Probes the head and the tail of the file, not all of it."""
    tokens = np.memmap(file_path, dtype=np.uint16, mode="r")
    n = len(tokens)
    top = int(tokens[:min(n, probe_tokens)].max()) if n else 0
    if n > probe_tokens:
        top = max(top, int(tokens[n - probe_tokens:].max()))
    if top >= vocab_size:
        raise ValueError(f"{file_path}: token id {top} does not fit vocab_size {vocab_size}; "
                         "the data was tokenised for another vocabulary")


def ensure_synthetic(data_dir: str, n_tokens: int, vocab_size: int, rank: int = 0, seed: int = 1729) -> str:
    """Create ``train.bin`` / ``val.bin`` under ``data_dir`` if missing or made for another vocabulary,
    size or seed (a sidecar ``synthetic.json`` records what the files hold). Returns ``data_dir``."""
    import json
    d = Path(data_dir)
    if rank == 0:
        d.mkdir(parents=True, exist_ok=True)
        want = {"vocab_size": int(vocab_size), "n_tokens": int(n_tokens), "seed": int(seed), "generator": 2}
        meta = d / "synthetic.json"
        try:
            have = json.loads(meta.read_text())
        except (OSError, ValueError):
            have = None
        for name, n, sd in (("train.bin", n_tokens, seed), ("val.bin", max(n_tokens // 10, 4096), seed + 1)):
            f = d / name
            if have != want or not f.exists() or f.stat().st_size != 2 * n:
                tmp = d / (name + f".tmp{os.getpid()}")
                synthetic_tokens(n, vocab_size, sd).tofile(tmp)
                os.replace(tmp, f)
        if have != want:
            tmp = d / f"synthetic.json.tmp{os.getpid()}"
            tmp.write_text(json.dumps(want))
            os.replace(tmp, meta)
    comm.barrier()
    return str(d)


def resolve_data_dir(dataset: str, data_dir: str = "") -> str:
    if data_dir:
        return data_dir
    here = Path(__file__).resolve().parent.parent
    return str(here / "data" / dataset)
