import pytest

from distributed_pytorch_amd.config import LLMconfig, Trainconfig, configs_from_args, safe_int_expr, build_parser


def test_safe_int_expr():
    assert safe_int_expr("2**14") == 16384
    assert safe_int_expr("16*1024") == 16384
    assert safe_int_expr("(1<<10) * 8 // 2") == 4096
    assert safe_int_expr("32768") == 32768
    for bad in ("__import__('os').system('true')", "2**1000", "a+1", "1.5", "open('x')"):
        with pytest.raises(Exception):
            safe_int_expr(bad)


def test_reference_flag_surface_present():
    """The reference's 35 flags (single-gpu/train.py:136-181) all parse."""
    flags = ("dataset batch_size max_iters eval_interval eval_iters learning_rate warmup_steps grad_clip act_recomp "
             "vocab_size block_size n_embd pos_emb n_layer dropout up_dim non_linearity n_exp n_shared n_act coeff "
             "alpha gamma attn n_head n_kv_heads q_latent_dim kv_latent_dim rope_head_dim total_batch_size_str moe "
             "aux_free eval save_model file_name").split()
    assert len(flags) == 35
    dests = {a.dest for a in build_parser()._actions}
    assert set(flags) <= dests


def test_effective_defaults_match_reference():
    tc, mc = configs_from_args([])
    # store_true flags default to False after the reference's override loop
    assert (mc.moe, mc.aux_free, tc.eval, tc.save_model, mc.act_recomp) == (False, False, False, False, False)
    assert (mc.n_layer, mc.n_embd, mc.attn, mc.pos_emb, mc.non_linearity, mc.up_dim) == (6, 256, "mla", "rope", "swiglu", 384)
    assert (mc.vocab_size, mc.block_size, tc.batch_size, tc.total_batch_size, tc.max_iters) == (50304, 1024, 2, 2048, 2500)
    assert (tc.learning_rate, tc.warmup_steps, tc.grad_clip) == (3e-4, 100, 1.0)


def test_override_and_fixups():
    tc, mc = configs_from_args("--attn MHA --n_head 12 --n_embd 768 --pos_emb Learn --non_linearity GELU "
                               "--total_batch_size_str 2**14 --moe --act_recomp --file_name MyRun".split())
    assert mc.attn == "mha" and mc.n_kv_heads == 12 and mc.pos_emb == "learn"
    assert mc.non_linearity == "GELU"           # case preserved like the reference
    assert tc.total_batch_size == 16384 and mc.moe and mc.act_recomp and tc.act_recomp
    assert tc.file_name == "MyRun"
    _, mq = configs_from_args("--attn mqa --n_head 8 --n_embd 64".split())
    assert mq.n_kv_heads == 1


def test_preset_and_explicit_override():
    tc, mc = configs_from_args("--preset gpt2-124m --n_layer 2".split())
    assert (mc.n_embd, mc.n_head, mc.up_dim, mc.attn, mc.pos_emb, mc.n_layer) == (768, 12, 3072, "mha", "learn", 2)


def test_mla_requires_latents():
    with pytest.raises(AssertionError):
        LLMconfig(attn="mla", q_latent_dim=None).finalize()


def test_own_communicator_refuses_non_default_stream_priorities():
    """A stream priority that stalls the own RCCL communicator after a few calls must not be selectable, and
    there is no environment override; channel bounds set before init_distributed() survive its own call of
    configure_rccl_env()."""
    import pytest
    from distributed_pytorch_amd.parallel import comm
    comm.check_own_priority("default")
    for p in ("high", "low"):
        with pytest.raises(ValueError, match="not usable under the own RCCL communicator"):
            comm.check_own_priority(p)
    saved = dict(comm._SETTINGS)
    try:
        comm.configure_rccl_env(max_channels=8, min_channels=2)
        comm.configure_rccl_env()            # what init_distributed() does
        assert comm._SETTINGS["max_ctas"] == 8 and comm._SETTINGS["min_ctas"] == 2
    finally:
        comm._SETTINGS.update(saved)
        import os
        os.environ.pop("NCCL_MAX_NCHANNELS", None)
        os.environ.pop("NCCL_MIN_NCHANNELS", None)
