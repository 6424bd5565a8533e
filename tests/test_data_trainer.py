import os

import numpy as np
import pytest
import torch

from distributed_pytorch_amd.config import LLMconfig, Trainconfig, configs_from_args
from distributed_pytorch_amd.data import DataLoader, ensure_synthetic, synthetic_tokens
from distributed_pytorch_amd.trainer import get_lr, train


def test_synthetic_tokens_shape():
    t = synthetic_tokens(50000, 50304)
    assert t.dtype == np.uint16 and t.max() < 50304 and (t == 50256).sum() > 50
    assert np.array_equal(t, synthetic_tokens(50000, 50304))
    # another seed is another sample of the SAME unigram distribution (what makes a validation loss on the
    # synthetic val split comparable with the training loss): the most frequent tokens are the same ones
    v = synthetic_tokens(50000, 50304, seed=1730)
    assert not np.array_equal(t, v)
    top_t = set(np.argsort(np.bincount(t, minlength=50304))[-20:].tolist())
    top_v = set(np.argsort(np.bincount(v, minlength=50304))[-20:].tolist())
    assert len(top_t & top_v) >= 15


def test_loader_semantics_and_rank_sharding(tmp_path):
    d = ensure_synthetic(str(tmp_path / "syn"), 20000, 1000)
    single = DataLoader(4, 16, os.path.join(d, "train.bin"), "cpu", 0, 1, seed=5)
    r0 = DataLoader(2, 16, os.path.join(d, "train.bin"), "cpu", 0, 2, seed=5)
    r1 = DataLoader(2, 16, os.path.join(d, "train.bin"), "cpu", 1, 2, seed=5)
    for _ in range(3):
        x, y = single.next_batch()
        assert x.shape == (4, 16) and x.dtype == torch.long
        assert torch.equal(x[:, 1:], y[:, :-1])        # y is x shifted by one
        x0, y0 = r0.next_batch()
        x1, y1 = r1.next_batch()
        assert torch.equal(torch.cat([x0, x1]), x) and torch.equal(torch.cat([y0, y1]), y)
    with pytest.raises(ValueError):
        DataLoader(4096, 16, os.path.join(d, "train.bin"), "cpu")


def test_lr_schedule_matches_reference_formula():
    tc = Trainconfig(learning_rate=1e-3, warmup_steps=10, max_iters=100)
    assert get_lr(0, tc) == pytest.approx(1e-4)
    assert get_lr(9, tc) == pytest.approx(1e-3)
    assert get_lr(10, tc) == pytest.approx(1e-3)
    assert get_lr(102, tc) == pytest.approx(1e-4)
    assert get_lr(500, tc) == pytest.approx(1e-4)
    mid = get_lr(56, tc)
    assert 1e-4 < mid < 1e-3


def test_single_trainer_cpu_checkpoint_loads_into_reference(tmp_path, reference_model_module, capsys):
    """BASELINE config 1: small GPT on the CPU through the single-GPU entry path; the checkpoint
    it writes loads into the reference LLM with strict=True."""
    import dataclasses
    argv = (f"--preset char-10m --n_layer 2 --n_embd 64 --n_head 4 --up_dim 128 --block_size 32 --vocab_size 300 "
            f"--dataset synthetic --synthetic_tokens 20000 --data_dir {tmp_path / 'd'} --batch_size 2 "
            f"--total_batch_size_str 2*32*2 --max_iters 6 --eval --eval_interval 3 --eval_iters 2 --warmup_steps 2 "
            f"--learning_rate 3e-3 --save_model --file_name {tmp_path / 'run'} --metrics_jsonl {tmp_path / 'm.jsonl'}").split()
    tc, mc = configs_from_args(argv)
    res = train(tc, mc, mode="single")
    out = capsys.readouterr().out
    assert "step: 0 | train loss:" in out and "grad_accum_steps: 2" in out and "val run" in out and "tok/s" in out
    assert res["losses"][-1] < res["losses"][0]
    ckpt = torch.load(str(tmp_path / "run_ckpt.pt"), weights_only=False)
    assert set(ckpt) >= {"model_config", "train_config", "model_state"}
    ref = reference_model_module
    names = [f.name for f in dataclasses.fields(ref.LLMconfig)]
    rm = ref.LLM(ref.LLMconfig(**{n: ckpt["model_config_dict"][n] for n in names}))
    rm.load_state_dict(ckpt["model_state"], strict=True)
    stats = torch.load(str(tmp_path / "run_stats.pt"), weights_only=False)
    assert stats["total_params"] == sum(p.numel() for p in rm.parameters())
    assert len(open(tmp_path / "m.jsonl").read().strip().splitlines()) == 7


def test_resume_continues_identically(tmp_path):
    base = (f"--n_layer 2 --n_embd 32 --n_head 4 --up_dim 64 --block_size 16 --vocab_size 128 --attn mha --pos_emb learn "
            f"--non_linearity gelu --dataset synthetic --synthetic_tokens 8000 --data_dir {tmp_path / 'd'} "
            f"--batch_size 2 --total_batch_size_str 64 --warmup_steps 2 --learning_rate 1e-2")
    tc, mc = configs_from_args((base + f" --max_iters 6 --file_name {tmp_path / 'a'}").split())
    full = train(tc, mc, "single")
    tc, mc = configs_from_args((base + f" --max_iters 6 --save_interval 4 --file_name {tmp_path / 'b'}").split())
    train(tc, mc, "single")
    tc, mc = configs_from_args((base + f" --max_iters 6 --resume {tmp_path / 'b_resume'} --file_name {tmp_path / 'c'}").split())
    resumed = train(tc, mc, "single")
    assert resumed["losses"] == pytest.approx(full["losses"][5:], rel=1e-5)
    # "--resume auto": the same command line picks up its own resume files when they exist ...
    tc, mc = configs_from_args((base + f" --max_iters 6 --save_interval 4 --resume auto --file_name {tmp_path / 'b'}").split())
    again = train(tc, mc, "single")
    assert again["losses"] == pytest.approx(full["losses"][5:], rel=1e-5)
    # ... and starts from scratch when there are none
    tc, mc = configs_from_args((base + f" --max_iters 6 --resume auto --file_name {tmp_path / 'fresh'}").split())
    fresh = train(tc, mc, "single")
    assert fresh["losses"] == pytest.approx(full["losses"], rel=1e-5)


def test_synthetic_data_is_rebuilt_for_another_vocabulary(tmp_path):
    """A token file made for a larger vocabulary must never reach a smaller model: the synthetic
    files are regenerated when the vocabulary changes, and check_vocab refuses a foreign file."""
    import numpy as np
    from distributed_pytorch_amd.data import check_vocab
    d = ensure_synthetic(str(tmp_path / "syn"), 20000, 50304)
    big = np.fromfile(os.path.join(d, "train.bin"), dtype=np.uint16)
    assert big.max() >= 1000
    with pytest.raises(ValueError):
        check_vocab(os.path.join(d, "train.bin"), 65)
    d = ensure_synthetic(str(tmp_path / "syn"), 20000, 65)
    small = np.fromfile(os.path.join(d, "train.bin"), dtype=np.uint16)
    assert small.max() < 65 and len(small) == 20000
    check_vocab(os.path.join(d, "train.bin"), 65)


def test_skipped_steps_are_counted():
    """A non-finite gradient norm skips the update; the engine counts it instead of hiding it."""
    import torch
    from distributed_pytorch_amd.config import LLMconfig
    from distributed_pytorch_amd.models.llm import LLM
    from distributed_pytorch_amd.parallel.engine import Engine
    torch.manual_seed(0)
    cfg = LLMconfig(vocab_size=64, block_size=8, n_embd=16, n_layer=1, n_head=2, up_dim=32, attn="mha",
                    pos_emb="learn", non_linearity="gelu").finalize()
    eng = Engine(LLM(cfg), dtype="fp32")
    x = torch.randint(0, 64, (2, 8))
    before = [o.master.clone() for o in eng.optims]
    eng.zero_grad()
    _, loss, _ = eng(x, x.roll(-1, dims=1))
    loss.backward()
    eng.groups[0].grad[0] = float("inf")
    eng.clip_grad_norm_(1.0)
    eng.step(1e-2)
    assert eng.skipped_steps() == 1
    assert all(torch.equal(a, o.master) for a, o in zip(before, eng.optims))
    eng.zero_grad()
    _, loss, _ = eng(x, x.roll(-1, dims=1))
    loss.backward()
    eng.clip_grad_norm_(1.0)
    eng.step(1e-2)
    assert eng.skipped_steps() == 1
    assert not torch.equal(before[0], eng.optims[0].master)


@pytest.mark.parametrize("flag,mode", [("--profile", "events"), ("--profile sync", "sync"), ("--profile kernels", "kernels"), ("", "")])
def test_profile_flag_and_phase_report(tmp_path, capsys, flag, mode):
    """``--profile`` alone means the non-synchronising event mode; every mode ends with a table of the
    forward / backward / optimizer phases (on the CPU the phases fall back to wall clock)."""
    argv = (f"--preset char-10m --n_layer 1 --n_embd 32 --n_head 2 --up_dim 64 --block_size 16 --vocab_size 128 "
            f"--dataset synthetic --synthetic_tokens 4000 --data_dir {tmp_path / 'd'} --batch_size 2 "
            f"--total_batch_size_str 2*16 --max_iters 4 --warmup_steps 1 --file_name {tmp_path / 'run'} {flag}").split()
    tc, mc = configs_from_args(argv)
    assert tc.profile == mode
    train(tc, mc, mode="single")
    out = capsys.readouterr().out
    if mode in ("events", "sync"):
        assert "phase (" in out and "forward" in out and "backward" in out and "optimizer" in out
    else:                                   # "kernels" prints the device's kernel table: nothing to show on the CPU
        assert "phase (" not in out
