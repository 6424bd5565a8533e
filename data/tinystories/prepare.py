"""TinyStories -> train.bin / val.bin (uint16 GPT-2-BPE ids with an end-of-text token after every
story, 99/1 split with seed 1729), written next to this script. Same output contract as the
reference (data/tinystories/prepare.py:13-52).

Needs the HuggingFace dataset ``roneneldan/TinyStories`` (cached or downloadable). Without it
(``--synthetic N`` or no network) a synthetic token stream of the TinyStories shape is written
instead: Zipf-distributed ids < 50304 with an EOT roughly every 250 tokens.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..", "..")))
from distributed_pytorch_amd.data import synthetic_tokens  # noqa: E402
from distributed_pytorch_amd.utils.tokenizer import get_encoder  # noqa: E402


def write_synthetic(n_tokens: int, out_dir: str):
    synthetic_tokens(n_tokens, 50304, 1729).tofile(os.path.join(out_dir, "train.bin"))
    synthetic_tokens(max(n_tokens // 100, 4096), 50304, 1730).tofile(os.path.join(out_dir, "val.bin"))
    print(f"wrote synthetic TinyStories-shaped streams: {n_tokens:,} train tokens")


def tokenize_and_save(out_dir: str = HERE, num_proc: int = 8, synthetic: int = 0):
    if synthetic:
        return write_synthetic(synthetic, out_dir)
    try:
        from datasets import load_dataset
        ds = load_dataset("roneneldan/TinyStories", split="train")
    except Exception as e:
        print(f"dataset unavailable ({e.__class__.__name__}); falling back to a synthetic stream")
        return write_synthetic(1 << 24, out_dir)
    enc = get_encoder("gpt2")
    split = ds.train_test_split(test_size=0.01, seed=1729, shuffle=True)
    split["val"] = split.pop("test")

    def process(example):
        ids = enc.encode_ordinary(example["text"])
        ids.append(enc.eot_token)
        return {"ids": ids, "len": len(ids)}

    tok = split.map(process, remove_columns=["text"], desc="tokenizing", num_proc=num_proc)
    for name, dset in tok.items():
        total = int(np.sum(dset["len"], dtype=np.uint64))
        arr = np.memmap(os.path.join(out_dir, f"{name}.bin"), dtype=np.uint16, mode="w+", shape=(total,))
        idx, shards = 0, 1024
        for i in range(shards):           # streamed write, never the whole corpus in memory
            batch = dset.shard(num_shards=shards, index=i, contiguous=True).with_format("numpy")
            chunk = np.concatenate(batch["ids"]) if len(batch) else np.zeros(0, np.uint16)
            arr[idx:idx + len(chunk)] = chunk
            idx += len(chunk)
        arr.flush()
        print(f"{name}: {total:,} tokens")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_proc", type=int, default=8)
    ap.add_argument("--synthetic", type=int, default=0, help="write a synthetic stream of this many tokens")
    a = ap.parse_args()
    tokenize_and_save(HERE, a.num_proc, a.synthetic)
