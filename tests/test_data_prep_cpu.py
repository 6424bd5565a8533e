"""The data preparation scripts run offline and write token files the loader accepts: uint16 ids, a
90/10 (Shakespeare) split, an end-of-text token in the TinyStories-shaped stream."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shakespeare_prepare_from_a_local_text(tmp_path):
    mod = _load(os.path.join(ROOT, "data", "shakespeare", "prepare.py"), "shakespeare_prepare")
    text = "\n".join(f"First Citizen {i}: speak, speak." for i in range(2000))
    src = tmp_path / "input.txt"
    src.write_text(text, encoding="utf-8")
    mod.tokenize_and_save(str(src), str(tmp_path), encoder="bytes")
    train = np.fromfile(tmp_path / "train.bin", dtype=np.uint16)
    val = np.fromfile(tmp_path / "val.bin", dtype=np.uint16)
    total = len(train) + len(val)
    assert total == len(text.encode("utf-8")) and abs(len(train) - 0.9 * total) <= 1
    assert int(train.max()) < 256                     # byte-level fallback ids
    # the files feed the loader and one training step
    from distributed_pytorch_amd.config import configs_from_args
    from distributed_pytorch_amd.trainer import train as run
    tc, mc = configs_from_args((f"--n_layer 1 --n_embd 32 --n_head 4 --up_dim 64 --block_size 16 --vocab_size 256 "
                                f"--attn mha --pos_emb learn --non_linearity gelu --dataset shakespeare --data_dir {tmp_path} "
                                f"--batch_size 2 --total_batch_size_str 32 --max_iters 1 --file_name {tmp_path / 'o'}").split())
    out = run(tc, mc, "single")
    assert np.isfinite(out["final_loss"])


def test_tinystories_prepare_synthetic_stream(tmp_path):
    mod = _load(os.path.join(ROOT, "data", "tinystories", "prepare.py"), "tinystories_prepare")
    mod.tokenize_and_save(str(tmp_path), num_proc=1, synthetic=50000)
    train = np.fromfile(tmp_path / "train.bin", dtype=np.uint16)
    val = np.fromfile(tmp_path / "val.bin", dtype=np.uint16)
    assert len(train) == 50000 and len(val) >= 4096
    assert int(train.max()) < 50304 and (train == 50256).sum() > 50      # GPT-2 end-of-text ids are present
