"""Tiny-Shakespeare -> train.bin / val.bin (uint16 token ids, 90/10 split), written next to
this script. Same output contract as the reference (data/shakespeare/prepare.py:7-36).

The text is read from ``input.txt`` beside this script (or ``--input``); if it is absent the
script tries to download it and, failing that (no network), generates a synthetic corpus of the
same size so that the training pipeline can still be exercised.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.normpath(os.path.join(HERE, "..", "..")))
from distributed_pytorch_amd.utils.tokenizer import get_encoder  # noqa: E402

URL = "https://raw.githubusercontent.com/karpathy/char-rnn/master/data/tinyshakespeare/input.txt"


def load_text(path: str) -> str:
    if os.path.exists(path):
        return open(path, "r", encoding="utf-8").read()
    try:
        import requests
        r = requests.get(URL, timeout=10)
        r.raise_for_status()
        with open(path, "w", encoding="utf-8") as f:
            f.write(r.text)
        return r.text
    except Exception as e:
        print(f"could not download the corpus ({e.__class__.__name__}); writing a synthetic stand-in")
        rng = np.random.default_rng(1729)
        words = ["the", "and", "to", "of", "thou", "my", "that", "in", "you", "is", "not", "with", "king", "lord",
                 "what", "shall", "love", "good", "come", "sir", "heart", "night", "death", "speak", "sweet"]
        lines = [" ".join(rng.choice(words, size=rng.integers(4, 11))).capitalize() + rng.choice([".", ",", "!", "?"])
                 for _ in range(40000)]
        return "\n".join(lines)


def tokenize_and_save(input_path: str = os.path.join(HERE, "input.txt"), out_dir: str = HERE, encoder: str = "gpt2"):
    text = load_text(input_path)
    enc = get_encoder(encoder)
    print(f"encoder: {getattr(enc, 'name_', getattr(enc, 'name', 'gpt2'))}")
    ids = np.array(enc.encode_ordinary(text), dtype=np.uint16)
    n = int(0.9 * len(ids))
    ids[:n].tofile(os.path.join(out_dir, "train.bin"))
    ids[n:].tofile(os.path.join(out_dir, "val.bin"))
    print(f"train has {n:,} tokens, val has {len(ids) - n:,} tokens")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", default=os.path.join(HERE, "input.txt"))
    ap.add_argument("--encoder", default="gpt2", choices=["gpt2", "bytes"])
    a = ap.parse_args()
    tokenize_and_save(a.input, HERE, a.encoder)
