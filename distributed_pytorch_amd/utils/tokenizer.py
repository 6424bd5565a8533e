"""Tokeniser selection for the data-preparation scripts.

The reference encodes with tiktoken's GPT-2 BPE (reference: data/shakespeare/prepare.py:20-24,
data/tinystories/prepare.py:27-31). On machines where tiktoken (or its vocabulary download) is
unavailable the scripts fall back, in order, to a local HuggingFace ``tokenizers`` GPT-2 file
(``GPT2_TOKENIZER_JSON``) and finally to a byte-level encoding (ids 0..255, EOT 256), so the
pipeline — uint16 ``train.bin`` / ``val.bin`` — always runs; the chosen encoder is printed.
"""
from __future__ import annotations

import os


class ByteTokenizer:
    name = "bytes"
    eot_token = 256
    n_vocab = 257

    def encode_ordinary(self, text: str):
        return list(text.encode("utf-8"))

    def decode(self, ids) -> str:
        return bytes(i for i in ids if 0 <= i < 256).decode("utf-8", errors="replace")


class _HFTokenizer:
    name = "hf-tokenizers-gpt2"

    def __init__(self, path: str):
        from tokenizers import Tokenizer
        self.tok = Tokenizer.from_file(path)
        self.eot_token = self.tok.token_to_id("<|endoftext|>") or 50256
        self.n_vocab = self.tok.get_vocab_size()

    def encode_ordinary(self, text: str):
        return self.tok.encode(text, add_special_tokens=False).ids

    def decode(self, ids) -> str:
        return self.tok.decode(list(ids))


def get_encoder(prefer: str = "gpt2"):
    if prefer == "gpt2":
        try:
            import tiktoken
            enc = tiktoken.get_encoding("gpt2")
            enc.encode_ordinary("probe")
            enc.name_ = "tiktoken-gpt2"
            return enc
        except Exception:
            pass
        path = os.environ.get("GPT2_TOKENIZER_JSON", "")
        if path and os.path.exists(path):
            try:
                return _HFTokenizer(path)
            except Exception:
                pass
    return ByteTokenizer()
