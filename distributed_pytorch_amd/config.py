"""Model / training configuration and the command-line surface.

Same field set and the same 35 flags as the reference trainers
(reference: single-gpu/train.py:29-206, single-gpu/model.py:39-75). Differences that are
deliberate:

* ``--total_batch_size_str`` is evaluated by a small arithmetic parser instead of ``eval``
  (``"2**14"`` and ``"16*1024"`` still work; arbitrary Python does not).
* Extra, optional flags (all default to the reference behaviour when absent) select the
  MI355X-specific machinery: ``--dtype``, ``--grad_dtype``, ``--bucket_mb``, ``--profile``, ``--hip_graph``,
  ``--resume``, ``--metrics_jsonl``, ``--synthetic_tokens`` and friends.
"""
from __future__ import annotations

import argparse
import ast
import operator
from dataclasses import asdict, dataclass, fields
from typing import Optional

__all__ = ["LLMconfig", "Trainconfig", "safe_int_expr", "build_parser", "configs_from_args", "default_configs",
           "PRESETS"]


@dataclass
class LLMconfig:
    # token params
    vocab_size: int = 50304
    block_size: int = 1024
    n_embd: int = 256
    pos_emb: str = "rope"            # 'learn' | 'sin' | 'rope'
    # feed-forward
    up_dim: int = 384
    non_linearity: str = "swiglu"
    dropout: float = 0.0
    n_layer: int = 6
    # MoE
    moe: bool = False
    n_exp: int = 16
    n_shared: int = 2
    n_act: int = 8                   # includes the shared experts
    coeff: float = 0.01
    aux_free: bool = False
    alpha: float = 0.0001            # complementary aux-loss coefficient (aux-free mode)
    gamma: float = 0.001             # bias update speed (aux-free mode)
    # attention
    attn: str = "mla"                # 'mha' | 'mqa' | 'gqa' | 'mla'
    n_head: int = 8
    n_kv_heads: int = 4
    q_latent_dim: Optional[int] = 32
    kv_latent_dim: Optional[int] = 32
    rope_head_dim: Optional[int] = 16
    act_recomp: bool = False
    act_recomp_scope: str = "attn"      # "attn": recompute attention only; "block": recompute whole blocks

    def finalize(self) -> "LLMconfig":
        """Apply the post-parse fix-ups of the reference (single-gpu/train.py:198-206)."""
        self.attn = self.attn.lower().strip()
        self.pos_emb = self.pos_emb.lower().strip()
        if self.attn == "mha":
            self.n_kv_heads = self.n_head
        elif self.attn == "mqa":
            self.n_kv_heads = 1
        elif self.attn == "gqa":
            assert self.n_head % self.n_kv_heads == 0, "n_head must be divisible by n_kv_heads"
        elif self.attn == "mla":
            assert self.q_latent_dim is not None and self.kv_latent_dim is not None, \
                "Either q_latent_dim or kv_latent_dim is missing"
            if self.pos_emb == "rope":
                assert self.rope_head_dim is not None, "Need dim of Rotary heads"
        else:
            raise ValueError(f"unknown attention type {self.attn!r}")
        if self.pos_emb not in ("learn", "sin", "rope"):
            raise ValueError(f"unknown positional embedding {self.pos_emb!r}")
        assert self.n_embd % self.n_head == 0, "n_embd must be divisible by n_head"
        return self

    def to_dict(self) -> dict:
        return asdict(self)


@dataclass
class Trainconfig:
    dataset: str = "tinystories"
    total_batch_size: int = 2 ** 11   # tokens per optimizer step (all ranks)
    batch_size: int = 2               # micro-batch (sequences per rank)
    max_iters: int = 2500
    eval: bool = False
    eval_interval: int = 100
    eval_iters: int = 100
    learning_rate: float = 3e-4
    warmup_steps: int = 100
    grad_clip: float = 1.0
    compile: bool = False             # accepted for parity; there is no tracing compiler here
    save_model: bool = False
    file_name: str = "llm_model"
    act_recomp: bool = False
    # ---- additions (not in the reference) ----
    weight_decay: float = 0.1
    dtype: str = "auto"               # 'auto' -> bf16 on GPU, fp32 on CPU; or 'bf16' / 'fp32'
    grad_dtype: str = "auto"          # dtype of the flat gradient buffer / gradient collectives
    bucket_mb: float = 48.0           # gradient bucket size for DDP / ZeRO
    fsdp_prefetch: int = 1            # FSDP: units all-gathered ahead of the running block (0: on demand)
    fsdp_inflight: int = 2            # FSDP: reduce-scatters in flight before backward waits for the oldest
    rccl_max_channels: int = 0        # NCCL_MAX_NCHANNELS of the RCCL communicator (0: RCCL's choice)
    rccl_min_channels: int = 0        # NCCL_MIN_NCHANNELS (0: RCCL's choice)
    comm_priority: str = "default"    # priority of the stream the collectives run on: default | high (c10d transport)
    vocab_parallel: bool = False      # split the head's vocabulary over the data-parallel ranks (ddp / zero / fsdp)
    ep_capacity: float = 0.0          # expert parallel: static capacity factor per peer (0 = exact, one host read / layer)
    tied_tail: str = "fused"          # tied lm_head/tkn_emb gradient: one all-reduce at the end | split (head part early)
    comm_backend: str = "c10d"        # GPU transport: c10d (ProcessGroupNCCL = RCCL) | own (csrc/comm communicator)
    data_dir: str = ""                # overrides ../data/<dataset>
    synthetic_tokens: int = 0         # >0: generate a synthetic uint16 token stream of that length
    seed: int = 1729
    resume: str = ""                  # resume checkpoint written with --save_interval; "auto" = this run's own, if present
    save_interval: int = 0
    save_async: bool = False          # resume files are written by a background thread and committed at the next save point
    metrics_jsonl: str = ""
    profile: str = ""                 # "" off | events (no synchronisation) | sync | kernels (per-kernel table)
    log_interval: int = 1
    sync_loss_every: int = 1          # host reads the loss every N steps (1 = reference behaviour)
    hip_graph: str = "auto"           # whole optimizer step replayed from one hipGraph: auto | on | off

    def to_dict(self) -> dict:
        return asdict(self)


# -------------------------------------------------------------------------------------------
_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.FloorDiv: operator.floordiv,
        ast.Pow: operator.pow, ast.LShift: operator.lshift, ast.Mod: operator.mod}


def safe_int_expr(text: str) -> int:
    """Evaluate an integer arithmetic expression (+ - * // ** << % and parentheses)."""
    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return node.value
        if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
            l, r = ev(node.left), ev(node.right)
            if isinstance(node.op, ast.Pow) and (abs(r) > 64 or abs(l) > 2 ** 32):
                raise ValueError("exponent too large")
            return _BIN[type(node.op)](l, r)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Div):
            l, r = ev(node.left), ev(node.right)
            q = l / r
            if q != int(q):
                raise ValueError("division is not exact")
            return int(q)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.UAdd, ast.USub)):
            v = ev(node.operand)
            return v if isinstance(node.op, ast.UAdd) else -v
        raise ValueError(f"unsupported expression element: {ast.dump(node)}")

    val = ev(ast.parse(str(text).strip(), mode="eval"))
    if val != int(val):
        raise ValueError(f"{text!r} is not an integer")
    return int(val)


def default_configs() -> tuple[Trainconfig, LLMconfig]:
    """The reference's module-level defaults (single-gpu/train.py:83-132). The boolean switches
    are False because the reference writes every parsed ``store_true`` flag back."""
    return Trainconfig(), LLMconfig()


# Named model shapes used by the benchmarks (reference state-dict schema, exact counts).
PRESETS = {
    "gpt2-124m": dict(n_layer=12, n_embd=768, n_head=12, up_dim=3072, attn="mha", pos_emb="learn",
                      non_linearity="gelu", vocab_size=50304, block_size=1024),
    "gpt2-350m": dict(n_layer=24, n_embd=1024, n_head=16, up_dim=4096, attn="mha", pos_emb="learn",
                      non_linearity="gelu", vocab_size=50304, block_size=1024),
    "gpt2-1.3b": dict(n_layer=24, n_embd=2048, n_head=16, up_dim=8192, attn="mha", pos_emb="learn",
                      non_linearity="gelu", vocab_size=50304, block_size=1024),
    "gpt-moe-8x": dict(n_layer=12, n_embd=768, n_head=12, up_dim=3072, attn="mha", pos_emb="learn",
                       non_linearity="gelu", vocab_size=50304, block_size=1024, moe=True, n_exp=8, n_shared=0,
                       n_act=2, aux_free=False),
    # the same with SwiGLU experts of 2/3 the width (equal parameter count): the reference's default activation
    "gpt-moe-8x-swiglu": dict(n_layer=12, n_embd=768, n_head=12, up_dim=2048, attn="mha", pos_emb="learn",
                              non_linearity="swiglu", vocab_size=50304, block_size=1024, moe=True, n_exp=8, n_shared=0,
                              n_act=2, aux_free=False),
    # a LLaMA-style 124M: grouped-query attention (4 KV heads), rotary positions, SwiGLU with 2/3 of the width
    "gqa-rope-swiglu-124m": dict(n_layer=12, n_embd=768, n_head=12, n_kv_heads=4, up_dim=2048, attn="gqa", pos_emb="rope",
                                 non_linearity="swiglu", vocab_size=50304, block_size=1024),
    "char-10m": dict(n_layer=6, n_embd=384, n_head=6, up_dim=1536, attn="mha", pos_emb="learn",
                     non_linearity="gelu", vocab_size=65, block_size=256),
}


def build_parser(tc: Trainconfig | None = None, mc: LLMconfig | None = None) -> argparse.ArgumentParser:
    tc = tc or Trainconfig()
    mc = mc or LLMconfig()
    p = argparse.ArgumentParser(description="Train a simple LLM model")
    # ---- the reference's 35 flags ----
    p.add_argument("--dataset", type=str, default=tc.dataset)
    p.add_argument("--batch_size", type=int, default=tc.batch_size)
    p.add_argument("--max_iters", type=int, default=tc.max_iters)
    p.add_argument("--eval_interval", type=int, default=tc.eval_interval)
    p.add_argument("--eval_iters", type=int, default=tc.eval_iters)
    p.add_argument("--learning_rate", type=float, default=tc.learning_rate)
    p.add_argument("--warmup_steps", type=int, default=tc.warmup_steps)
    p.add_argument("--grad_clip", type=float, default=tc.grad_clip)
    p.add_argument("--act_recomp", action="store_true")
    p.add_argument("--act_recomp_scope", type=str, default=mc.act_recomp_scope, choices=["attn", "block"])
    p.add_argument("--vocab_size", type=int, default=mc.vocab_size)
    p.add_argument("--block_size", type=int, default=mc.block_size)
    p.add_argument("--n_embd", type=int, default=mc.n_embd)
    p.add_argument("--pos_emb", type=str, default=mc.pos_emb)
    p.add_argument("--n_layer", type=int, default=mc.n_layer)
    p.add_argument("--dropout", type=float, default=mc.dropout)
    p.add_argument("--up_dim", type=int, default=mc.up_dim)
    p.add_argument("--non_linearity", type=str, default=mc.non_linearity)
    p.add_argument("--n_exp", type=int, default=mc.n_exp)
    p.add_argument("--n_shared", type=int, default=mc.n_shared)
    p.add_argument("--n_act", type=int, default=mc.n_act)
    p.add_argument("--coeff", type=float, default=mc.coeff)
    p.add_argument("--alpha", type=float, default=mc.alpha)
    p.add_argument("--gamma", type=float, default=mc.gamma)
    p.add_argument("--attn", type=str, default=mc.attn)
    p.add_argument("--n_head", type=int, default=mc.n_head)
    p.add_argument("--n_kv_heads", type=int, default=mc.n_kv_heads)
    p.add_argument("--q_latent_dim", type=int, default=mc.q_latent_dim)
    p.add_argument("--kv_latent_dim", type=int, default=mc.kv_latent_dim)
    p.add_argument("--rope_head_dim", type=int, default=mc.rope_head_dim)
    p.add_argument("--total_batch_size_str", type=str, default=str(tc.total_batch_size))
    p.add_argument("--moe", action="store_true")
    p.add_argument("--aux_free", action="store_true")
    p.add_argument("--eval", action="store_true")
    p.add_argument("--save_model", action="store_true")
    p.add_argument("--file_name", type=str, default=tc.file_name)
    # ---- additions ----
    p.add_argument("--preset", type=str, default="", choices=[""] + sorted(PRESETS),
                   help="named model shape; explicit model flags still override it")
    p.add_argument("--weight_decay", type=float, default=tc.weight_decay)
    p.add_argument("--dtype", type=str, default=tc.dtype, choices=["auto", "bf16", "fp32"],
                   help="compute dtype of the parameters and activations (auto: bf16 on the GPU, fp32 on the CPU); the optimizer keeps fp32 master weights")
    p.add_argument("--grad_dtype", type=str, default=tc.grad_dtype, choices=["auto", "bf16", "fp32"],
                   help="dtype of the flat gradient buffers and of the gradient collectives")
    p.add_argument("--bucket_mb", type=float, default=tc.bucket_mb, help="size of a gradient bucket (one collective) in MiB")
    p.add_argument("--fsdp_prefetch", type=int, default=tc.fsdp_prefetch, help="FSDP: units all-gathered ahead of the running block")
    p.add_argument("--fsdp_inflight", type=int, default=tc.fsdp_inflight, help="FSDP: reduce-scatters in flight before backward waits")
    p.add_argument("--rccl_max_channels", type=int, default=tc.rccl_max_channels, help="upper bound on RCCL channels, i.e. CUs a collective takes (0: RCCL decides)")
    p.add_argument("--rccl_min_channels", type=int, default=tc.rccl_min_channels, help="lower bound on RCCL channels (0: RCCL decides)")
    p.add_argument("--comm_priority", type=str, default=tc.comm_priority, choices=["default", "high"],
                   help="priority of the HIP stream the collectives run on (high: c10d transport only)")
    p.add_argument("--vocab_parallel", action="store_true", default=tc.vocab_parallel,
                   help="split the head's vocabulary over the ranks (pays below vocabulary / ranks tokens per rank and step)")
    p.add_argument("--ep_capacity", type=float, default=tc.ep_capacity,
                   help="expert parallel: static per-peer capacity factor, no host read per layer, overflow pairs dropped (0: exact variable all-to-all)")
    p.add_argument("--tied_tail", type=str, default=tc.tied_tail, choices=["fused", "split"],
                   help="DDP: reduce the tied embedding / head gradient as one tensor at the end of backward, or the head's part early")
    p.add_argument("--comm_backend", type=str, default=tc.comm_backend, choices=["c10d", "own"],
                   help="GPU transport: torch.distributed's RCCL process group, or the framework's own RCCL communicator")
    p.add_argument("--data_dir", type=str, default=tc.data_dir, help="directory of train.bin / val.bin (default: data/<dataset>)")
    p.add_argument("--synthetic_tokens", type=str, default=str(tc.synthetic_tokens), help="--dataset synthetic: tokens to generate (an arithmetic expression is accepted)")
    p.add_argument("--seed", type=int, default=tc.seed)
    p.add_argument("--resume", type=str, default=tc.resume,
                   help="path prefix of resume files to continue from, or 'auto': this run's own files if a complete generation exists (the job may have another number of ranks than the one that wrote them)")
    p.add_argument("--save_interval", type=int, default=tc.save_interval, help="write resume files (optimizer shards, RNG, loader state) every N iterations (0: never)")
    p.add_argument("--save_async", action="store_true", default=tc.save_async,
                   help="write the periodic resume files in the background (snapshot to pinned host memory, committed once every rank's file is complete)")
    p.add_argument("--metrics_jsonl", type=str, default=tc.metrics_jsonl, help="append one JSON record per step (loss, step time, tokens/s, MFU, lr, memory) to this file")
    p.add_argument("--profile", type=str, nargs="?", const="events", default=tc.profile,
                   choices=["", "events", "sync", "kernels"],
                   help="phase timing from stream events (default), synchronised wall clock, or a per-kernel table")
    p.add_argument("--log_interval", type=int, default=tc.log_interval, help="print the step line every N iterations")
    p.add_argument("--sync_loss_every", type=int, default=tc.sync_loss_every, help="read the loss back to the host every N steps (each read drains the GPU queue)")
    p.add_argument("--hip_graph", type=str, nargs="?", const="on", default=tc.hip_graph, choices=["auto", "on", "off"],
                   help="replay the whole optimizer step from one hipGraph (single GPU); auto = launch-bound steps only")
    return p


def configs_from_args(argv=None, tc: Trainconfig | None = None, mc: LLMconfig | None = None
                      ) -> tuple[Trainconfig, LLMconfig]:
    """Parse ``argv`` and distribute every value to the training or the model config the way the
    reference's override loop does (single-gpu/train.py:183-197): a key goes to the training
    config if it has that attribute, otherwise to the model config; strings are lower-cased
    except ``non_linearity``; ``act_recomp`` goes to both."""
    tc = tc or Trainconfig()
    mc = mc or LLMconfig()
    parser = build_parser(tc, mc)
    args = parser.parse_args(argv)
    given = set()
    if argv is not None or True:
        import sys
        raw = list(sys.argv[1:] if argv is None else argv)
        given = {a.split("=")[0].lstrip("-") for a in raw if a.startswith("--")}
    if args.preset:
        for k, v in PRESETS[args.preset].items():
            if k not in given:
                setattr(mc, k, v)
    tnames = {f.name for f in fields(Trainconfig)}
    mnames = {f.name for f in fields(LLMconfig)}
    for key, value in vars(args).items():
        if key == "preset":
            continue
        if args.preset and key in PRESETS[args.preset] and key not in given:
            continue
        if key == "total_batch_size_str":
            tc.total_batch_size = safe_int_expr(value)
            continue
        if key == "synthetic_tokens":
            tc.synthetic_tokens = safe_int_expr(value)
            continue
        if key == "act_recomp":
            mc.act_recomp = bool(value)
            tc.act_recomp = bool(value)
            continue
        if isinstance(value, str) and key not in ("non_linearity", "file_name", "data_dir", "resume", "metrics_jsonl"):
            value = value.lower().strip()
        if key in tnames:
            setattr(tc, key, value)
        elif key in mnames:
            setattr(mc, key, value)
    mc.finalize()
    return tc, mc
