"""The configurable decoder-only LLM.

Same public surface and the same ``state_dict`` key schema as the reference model
(reference: single-gpu/model.py:98-747): ``LLM(config)``, ``forward(idx, targets=None,
kv_caches=None) -> (logits, loss, kv_caches)``, ``generate``, ``configure_optimizers``,
``get_num_params``, attribute ``VAL_RUN``; attention in {mha, mqa, gqa, mla}, positions in
{learn, sin, rope}, dense MLP or DeepSeek-style MoE, KV-cache decoding, activation
recomputation. A checkpoint written from this model loads into the reference ``LLM`` with
``strict=True`` and vice versa.

What is different underneath:

* every hot op goes through ``ops`` (HIP kernels on the GPU): fused embedding gather,
  LayerNorm with fused residual add, flash attention that reads the fused QKV projection in
  place and is GQA-native, exact-erf GELU / SwiGLU kernels, fused lm_head + cross-entropy that
  never materialises the logits;
* the compute dtype is the parameter dtype (the engines in ``parallel/`` keep bf16 compute
  copies of fp32 master weights in flat buffers); there is no autocast;
* the residual stream is carried as ``(x, delta)`` so that each residual add is fused into
  the next LayerNorm;
* MLA is expressed as multi-query flash attention over the latent KV (absorbed projections)
  instead of materialising ``(B, nh, T, T)`` scores;
* MoE routes with a fused top-k gate kernel and runs the routed experts as grouped GEMMs over an
  expert-sorted, block-padded token layout built on the device (no per-expert ``nonzero`` host
  round trips, no expert loop); the aux-free bias is updated exactly once per training forward,
  also under activation recomputation; attention dropout runs inside the flash kernels;
* cached decoding uses a bottom-right aligned causal mask (the reference's GQA path attends to
  key 0 only when decoding from a cache).
"""
from __future__ import annotations

import contextlib
import math
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from .. import ops
from ..config import LLMconfig

__all__ = ["LLM", "Block", "GQA", "NaiveMHLA", "FullMHLA", "Attention", "MLP", "Expert", "MoE"]

_tls = threading.local()


def _in_recompute() -> bool:
    return getattr(_tls, "recompute", False)


@contextlib.contextmanager
def _recompute_scope():
    prev = getattr(_tls, "recompute", False)
    _tls.recompute = True
    try:
        yield
    finally:
        _tls.recompute = prev


def _ckpt_contexts():
    return contextlib.nullcontext(), _recompute_scope()


def _drop(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    return F.dropout(x, p, training) if (p > 0.0 and training) else x


# ---------------------------------------------------------------------------------------------
class StaticKVCache:
    """Preallocated key / value cache for decoding inside a captured hipGraph: ``k[i], v[i]
    [B, S, heads, width]`` per layer, and the fill level as DEVICE scalars (``pos`` int64 = position of
    the next token, ``len_after`` int32 = fill level once that token is appended), so a replayed
    step needs no host-side shape or offset. ``advance()`` moves both by one.

    GQA family: ``heads = n_kv_heads``, ``width = head_size``. MLA: one latent "head"; a key row is the
    latent (plus the rotary key for the RoPE variant), a value row the latent, both zero-padded to
    the kernels' 64 / 128 columns (``static_width``)."""

    @staticmethod
    def static_width(config: LLMconfig) -> int | None:
        """Row width of the cache, or None if this configuration has no static-cache decode path."""
        if config.attn in ("mha", "mqa", "gqa"):
            hs = config.n_embd // config.n_head
            return hs if hs in (64, 128) else None
        dk = config.kv_latent_dim + (config.rope_head_dim if config.pos_emb == "rope" else 0)
        return 64 if dk <= 64 else (128 if dk <= 128 else None)

    def __init__(self, config: LLMconfig, batch: int, max_len: int, device, dtype):
        self.mla = config.attn == "mla"
        self.width = self.static_width(config)
        assert self.width is not None, "no static-cache decode path for this configuration"
        shape = (batch, max_len, 1 if self.mla else config.n_kv_heads, self.width)
        self.k = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(config.n_layer)]
        self.v = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(config.n_layer)]
        self.max_len = max_len
        self.pos = torch.zeros(1, dtype=torch.int64, device=device)
        self.len_after = torch.ones(1, dtype=torch.int32, device=device)

    def fill(self, caches, length: int):
        """Load the prompt's keys / values from the dynamic caches an eager forward returned."""
        for i, c in enumerate(caches):
            if not self.mla:
                ck, cv = c                                           # (B, nkv, T, hs)
                self.k[i][:, :length].copy_(ck.transpose(1, 2))
                self.v[i][:, :length].copy_(cv.transpose(1, 2))
                continue
            if isinstance(c, dict):                                  # RoPE variant: latent + rotary key
                lat, kr = c["c_kv"], c["k_r"].squeeze(1)             # (B, T, n_kvl), (B, T, dhr)
                self.k[i][:, :length, 0, : lat.shape[-1]].copy_(lat)
                self.k[i][:, :length, 0, lat.shape[-1]: lat.shape[-1] + kr.shape[-1]].copy_(kr)
            else:
                lat = c
                self.k[i][:, :length, 0, : lat.shape[-1]].copy_(lat)
            self.v[i][:, :length, 0, : lat.shape[-1]].copy_(lat)
        self.pos.fill_(length)
        self.len_after.fill_(length + 1)

    def advance(self):
        self.pos.add_(1)
        self.len_after.add_(1)

    def layer(self, i: int):
        return _StaticLayer(self.k[i], self.v[i], self.pos, self.len_after)


class _StaticLayer:
    __slots__ = ("k", "v", "pos", "len_after")

    def __init__(self, k, v, pos, len_after):
        self.k, self.v, self.pos, self.len_after = k, v, pos, len_after


# ---------------------------------------------------------------------------------------------
class GQA(nn.Module):
    """Grouped-query attention (MHA / MQA / GQA) with or without RoPE
    (reference: single-gpu/model.py:98-155)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        if config.attn == "mha":
            config.n_kv_heads = config.n_head
        elif config.attn == "mqa":
            config.n_kv_heads = 1
        else:
            assert config.n_head % config.n_kv_heads == 0, "n_head must be divisible by n_kv_heads"
        assert config.n_embd % config.n_head == 0, "n_embd must be divisible by n_head"
        self.config = config
        self.head_size = config.n_embd // config.n_head
        self.c_attn = nn.Linear(config.n_embd, config.n_embd + 2 * config.n_kv_heads * self.head_size)
        self.c_proj = nn.Linear(config.n_embd, config.n_embd)
        self.attn_dropout = nn.Dropout(config.dropout)   # kept for module-tree parity; unused like the reference's
        self.resid_dropout = nn.Dropout(config.dropout)

    def forward(self, x, rope=None, kv_cache=None, VAL_RUN=False, residual=None):
        """``residual`` (the block's residual stream) is added to the output projection inside its GEMM;
        the result is then the updated stream, not the attention branch alone."""
        B, T, C = x.shape
        nh, nkv, hs = self.config.n_head, self.config.n_kv_heads, self.head_size
        p_drop = self.config.dropout if self.training else 0.0
        if (rope is not None and kv_cache is None and not self._want_cache()
                and ops.qkv_rope_attention_supported(x, self.c_attn.weight, self.c_attn.bias, rope[0], rope[1], int(rope[2]), hs)):
            # training path: the rotation is the projection GEMM's epilogue, its inverse on dq / dk the attention
            # backward kernels' store; no rotary pass over the activations in either direction
            y = ops.qkv_rope_attention(x, self.c_attn.weight, self.c_attn.bias, rope[0], rope[1], int(rope[2]), nh, nkv,
                                       dropout_p=p_drop, sole_use=True).reshape(B, T, C)
            return self._project(y, residual), None
        qkv = ops.linear(x, self.c_attn.weight, self.c_attn.bias, sole_use=True)
        if rope is not None:
            # rotate the q and k heads (contiguous leading part of the fused projection)
            cos_t, sin_t, pos0 = rope
            qkv = ops.rope_qkv_(qkv, cos_t, sin_t, pos0, nh + nkv, hs)   # in place: no copy, no re-concatenation
        if isinstance(kv_cache, _StaticLayer):
            # one new token appended in place at the device-side position, attention over the filled part
            q, k, v = ops.split_qkv(qkv, nh, nkv)
            ops.kv_append(kv_cache.k, kv_cache.v, k, v, kv_cache.pos)
            y = ops.cached_attention(q, kv_cache.k, kv_cache.v, kv_cache.len_after)
            new_cache = kv_cache
        elif kv_cache is None and not self._want_cache():
            y = ops.causal_self_attention_qkv(qkv, nh, nkv, dropout_p=p_drop)
            new_cache = None
        else:
            q, k, v = ops.split_qkv(qkv, nh, nkv)
            if kv_cache is not None:
                past_k, past_v = kv_cache                      # (B, nkv, Tp, hs) views of (B, Tp, nkv, hs)
                k = torch.cat((past_k.transpose(1, 2), k), dim=1)
                v = torch.cat((past_v.transpose(1, 2), v), dim=1)
            y = ops.causal_attention(q, k, v, dropout_p=p_drop)
            new_cache = (k.transpose(1, 2), v.transpose(1, 2))  # reference cache shape (B, nkv, T, hs)
        return self._project(y.reshape(B, T, C), residual), new_cache

    def _project(self, y, residual):
        if residual is not None and not (self.training and self.config.dropout > 0.0):
            return ops.linear(y, self.c_proj.weight, self.c_proj.bias, sole_use=True, residual=residual)
        y = _drop(ops.linear(y, self.c_proj.weight, self.c_proj.bias, sole_use=True), self.config.dropout, self.training)
        return y if residual is None else residual + y

    def _want_cache(self) -> bool:
        # The training / validation path never reads the returned cache; building it would only
        # cost memory. Inference (eval mode outside a validation run) returns it.
        return not self.training and not getattr(self, "_val_run", False)


def _mla_static_step(cache, q_eff, k_row, v_row, scale):
    """One decode step of latent attention over a static cache: q_eff [B, 1, nh, Dk], the new key row
    k_row [B, 1, Dk] and value row v_row [B, 1, Dv] are zero-padded to the cache width, appended at the
    device-side position, and the context [B, 1, nh, Dv] is read back from the padded output."""
    W = cache.k.shape[-1]
    B = q_eff.shape[0]
    dk, dv = k_row.shape[-1], v_row.shape[-1]
    ops.kv_append(cache.k, cache.v, F.pad(k_row, (0, W - dk)).view(B, 1, 1, W),
                  F.pad(v_row, (0, W - dv)).view(B, 1, 1, W), cache.pos)
    ctx = ops.cached_attention(F.pad(q_eff, (0, W - dk)), cache.k, cache.v, cache.len_after, scale)
    return ctx[..., :dv]


def _mla_attention(q_eff, k_eff, v_lat, scale, dropout_p):
    """Multi-query attention over latent keys/values (the absorbed form, used with KV caches): q_eff [B,T,nh,Dk],
    k_eff [B,Tk,1,Dk], v_lat [B,Tk,1,Dv] -> [B,T,nh,Dv]. Widths the flash kernels are built for (zero-padded up
    inside ``ops.causal_attention``) run on them, wider ones on the eager path."""
    return ops.causal_attention(q_eff, k_eff, v_lat, scale, dropout_p)


class NaiveMHLA(nn.Module):
    """Multi-head latent attention without RoPE (reference: single-gpu/model.py:157-235)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        assert config.n_embd % config.n_head == 0, "num of heads must be a divisor of n_embd"
        self.head_size = config.n_embd // config.n_head
        self.config = config
        self.W_dq = nn.Linear(config.n_embd, config.q_latent_dim, bias=False)
        self.W_uq = nn.Linear(config.q_latent_dim, config.n_embd, bias=False)
        self.W_dkv = nn.Linear(config.n_embd, config.kv_latent_dim, bias=False)
        self.W_uk = nn.Linear(config.kv_latent_dim, config.n_embd, bias=False)
        self.W_uv = nn.Linear(config.kv_latent_dim, config.n_embd, bias=False)
        self.W_o = nn.Linear(config.n_embd, config.n_embd, bias=False)
        self.dropout = nn.Dropout(config.dropout)
        self.register_buffer("_k_absorbed_inference", None)
        self.register_buffer("_v_absorbed_inference", None)

    def _absorbed(self):
        nh, n_kvl, hs = self.config.n_head, self.config.kv_latent_dim, self.head_size
        # Same absorbed products as the reference (single-gpu/model.py:196-197), including its choice
        # of folding W_dq^T W_uq^T into the key side: (nh, hs, n_kvl) and (nh, n_kvl, hs).
        k_abs = (self.W_dq.weight.t() @ self.W_uq.weight.t() @ self.W_uk.weight).view(nh, hs, n_kvl)
        v_abs = (self.W_uv.weight.t() @ self.W_o.weight.t()).view(n_kvl, nh, hs).transpose(0, 1)
        return k_abs, v_abs

    def forward(self, x, rope=None, kv_cache=None, VAL_RUN=False):
        B, T, C = x.shape
        nh, n_kvl, hs = self.config.n_head, self.config.kv_latent_dim, self.head_size
        p_drop = self.config.dropout if self.training else 0.0
        new_c_kv = ops.linear(x, self.W_dkv.weight)
        q = ops.linear(ops.linear(x, self.W_dq.weight), self.W_uq.weight).view(B, T, nh, hs)
        if kv_cache is None and (self.training or VAL_RUN):
            # Per-head form of the same function (no latent-wide scores): the absorbed key matrix is applied to the
            # latent rows instead of the queries, K = c_kv (W_dq^T W_uq^T W_uk)^T, and the value / output products
            # to the latent rows instead of the context, V = (c_kv W_uv^T) W_o^T. Scores and context are then
            # ordinary hs-wide heads, which the flash kernels take; FLOPs per (query, key) pair drop from
            # 2 n_kvl to 2 hs.
            w_k = self.W_dq.weight.t() @ self.W_uq.weight.t() @ self.W_uk.weight          # (C, n_kvl): k_abs, not re-viewed
            k_heads = ops.linear(new_c_kv, w_k.to(q.dtype)).view(B, T, nh, hs)
            v_heads = ops.linear(ops.linear(new_c_kv, self.W_uv.weight), self.W_o.weight).view(B, T, nh, hs)
            y = ops.causal_attention(q, k_heads, v_heads, 1.0 / math.sqrt(hs), p_drop).reshape(B, T, C)
            y = _drop(y, self.config.dropout, self.training)
            return y, new_c_kv
        if self.training or VAL_RUN:
            k_abs, v_abs = self._absorbed()
        else:
            if self._k_absorbed_inference is None or self._v_absorbed_inference is None:
                with torch.no_grad():
                    k_abs, v_abs = self._absorbed()
                    self._k_absorbed_inference, self._v_absorbed_inference = k_abs.clone(), v_abs.clone()
            k_abs, v_abs = self._k_absorbed_inference, self._v_absorbed_inference
        q_eff = torch.einsum("bthd,hdl->bthl", q, k_abs.to(q.dtype))          # (B, T, nh, n_kvl)
        if isinstance(kv_cache, _StaticLayer):
            ctx = _mla_static_step(kv_cache, q_eff, new_c_kv, new_c_kv, 1.0 / math.sqrt(hs))
            y = torch.einsum("bthl,hld->bthd", ctx, v_abs.to(ctx.dtype)).reshape(B, T, C)
            return y, kv_cache
        c_kv = new_c_kv if kv_cache is None else torch.cat([kv_cache, new_c_kv], dim=1)
        ctx = _mla_attention(q_eff, c_kv.unsqueeze(2), c_kv.unsqueeze(2), 1.0 / math.sqrt(hs), p_drop)
        y = torch.einsum("bthl,hld->bthd", ctx, v_abs.to(ctx.dtype)).reshape(B, T, C)
        y = _drop(y, self.config.dropout, self.training)
        return y, c_kv


class FullMHLA(nn.Module):
    """Multi-head latent attention with decoupled RoPE, DeepSeek-V2 style
    (reference: single-gpu/model.py:237-345)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        assert config.n_embd % config.n_head == 0, "num of heads must be a divisor of n_embd"
        self.config = config
        self.W_dq = nn.Linear(config.n_embd, config.q_latent_dim, False)
        self.dropout = nn.Dropout(config.dropout)
        self.head_size = config.n_embd // config.n_head
        self.W_uq = nn.Linear(config.q_latent_dim, config.n_embd, False)
        self.W_dkv = nn.Linear(config.n_embd, config.kv_latent_dim, False)
        self.W_uk = nn.Linear(config.kv_latent_dim, config.n_embd, False)
        self.W_uv = nn.Linear(config.kv_latent_dim, config.n_embd, False)
        self.W_qr = nn.Linear(config.q_latent_dim, config.n_head * config.rope_head_dim, False)
        self.W_kr = nn.Linear(config.n_embd, config.rope_head_dim, False)
        self.W_o = nn.Linear(config.n_embd, config.n_embd, False)
        self.register_buffer("_k_absorbed_inference", None, persistent=False)
        self.register_buffer("_v_absorbed_inference", None, persistent=False)

    def _absorbed(self):
        nh, nlkv, hs, nlq = self.config.n_head, self.config.kv_latent_dim, self.head_size, self.config.q_latent_dim
        # Same absorbed products as the reference (single-gpu/model.py:291-292): W_uq.weight is
        # re-viewed (not transposed) as (nlq, nh, hs), exactly as the reference does, so that a
        # checkpoint means the same function in both code bases. (nh, nlq, nlkv) and (nh, nlkv, hs).
        k_abs = self.W_uq.weight.view(nlq, nh, hs).transpose(0, 1) @ self.W_uk.weight.view(nh, hs, nlkv)
        v_abs = (self.W_uv.weight.t() @ self.W_o.weight.t()).view(nlkv, nh, hs).transpose(0, 1)
        return k_abs, v_abs

    def forward(self, x, rope=None, kv_cache=None, VAL_RUN=False):
        B, T, C = x.shape
        nh, nlkv = self.config.n_head, self.config.kv_latent_dim
        hs, dhr = self.head_size, self.config.rope_head_dim
        p_drop = self.config.dropout if self.training else 0.0
        cos_t, sin_t, pos0 = rope
        c_q = ops.linear(x, self.W_dq.weight)                                   # (B, T, nlq)
        new_c_kv = ops.linear(x, self.W_dkv.weight)
        k_r = ops.apply_rope(ops.linear(x, self.W_kr.weight).view(B, T, 1, dhr), cos_t, sin_t, pos0)
        q_r = ops.apply_rope(ops.linear(c_q, self.W_qr.weight).view(B, T, nh, dhr), cos_t, sin_t, pos0)
        if kv_cache is None and (self.training or VAL_RUN):
            # Per-head form of the same function: q_nope . c_kv = (c_q A_h) . (W_uk_h c_kv) with A = W_uq.weight
            # re-viewed as (nlq, C), so the scores are [Q_h | q_r] . [K_h | k_r], hs + dhr wide instead of
            # nlkv + dhr, and V = (c_kv W_uv^T) W_o^T carries the output product, hs wide instead of nlkv.
            # (key, value) widths (256, 128) for the 1024 / 8 heads / rope 128 configuration: flash kernels.
            nlq = self.config.q_latent_dim
            q_heads = ops.linear(c_q, self.W_uq.weight.view(nlq, C).t().contiguous()).view(B, T, nh, hs)
            k_heads = ops.linear(new_c_kv, self.W_uk.weight).view(B, T, nh, hs)
            v_heads = ops.linear(ops.linear(new_c_kv, self.W_uv.weight), self.W_o.weight).view(B, T, nh, hs)
            q_full = torch.cat([q_heads, q_r], dim=-1)
            k_full = torch.cat([k_heads, k_r.expand(B, T, nh, dhr)], dim=-1)
            y = ops.causal_attention(q_full, k_full, v_heads, 1.0 / math.sqrt(hs + dhr), p_drop).reshape(B, T, C)
            y = _drop(y, self.config.dropout, self.training)
            return y, {"c_kv": new_c_kv, "k_r": k_r.transpose(1, 2)}
        if self.training or VAL_RUN:
            k_abs, v_abs = self._absorbed()
        else:
            if self._k_absorbed_inference is None or self._v_absorbed_inference is None:
                with torch.no_grad():
                    k_abs, v_abs = self._absorbed()
                    self._k_absorbed_inference, self._v_absorbed_inference = k_abs.clone(), v_abs.clone()
            k_abs, v_abs = self._k_absorbed_inference, self._v_absorbed_inference
        q_nope = torch.einsum("btq,hql->bthl", c_q, k_abs.to(c_q.dtype))       # (B, T, nh, nlkv)
        q_eff = torch.cat([q_nope, q_r], dim=-1)                               # (B, T, nh, nlkv + dhr)
        if isinstance(kv_cache, _StaticLayer):
            k_row = torch.cat([new_c_kv, k_r.view(B, T, dhr)], dim=-1)
            ctx = _mla_static_step(kv_cache, q_eff, k_row, new_c_kv, 1.0 / math.sqrt(hs + dhr))
            y = torch.einsum("bthl,hld->bthd", ctx, v_abs.to(ctx.dtype)).reshape(B, T, C)
            return y, kv_cache
        c_kv = new_c_kv if kv_cache is None else torch.cat([kv_cache["c_kv"], new_c_kv], dim=1)
        k_r = k_r.transpose(1, 2)                                              # (B, 1, T, dhr): cache layout
        if kv_cache is not None:
            k_r = torch.cat([kv_cache["k_r"], k_r], dim=2)
        k_eff = torch.cat([c_kv, k_r.squeeze(1)], dim=-1).unsqueeze(2)         # (B, Tk, 1, nlkv + dhr)
        ctx = _mla_attention(q_eff, k_eff, c_kv.unsqueeze(2), 1.0 / math.sqrt(hs + dhr), p_drop)
        y = torch.einsum("bthl,hld->bthd", ctx, v_abs.to(ctx.dtype)).reshape(B, T, C)
        y = _drop(y, self.config.dropout, self.training)
        return y, {"c_kv": c_kv, "k_r": k_r}


class Attention(nn.Module):
    """Routes to the configured attention (reference: single-gpu/model.py:347-363)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        self.config = config
        if config.attn in ("mha", "mqa", "gqa"):
            self.attn = GQA(config)
        elif config.attn == "mla":
            self.attn = NaiveMHLA(config) if config.pos_emb != "rope" else FullMHLA(config)
        else:
            raise ValueError(f"unknown attention {config.attn!r}")

    def forward(self, x, rope=None, kv_cache=None, VAL_RUN=False, residual=None):
        if isinstance(self.attn, GQA):
            self.attn._val_run = VAL_RUN
            return self.attn(x, rope, kv_cache, VAL_RUN, residual)
        y, new_cache = self.attn(x, rope, kv_cache, VAL_RUN)
        return (y if residual is None else residual + y), new_cache


# ---------------------------------------------------------------------------------------------
class MLP(nn.Module):
    """Feed-forward block, bias-free (reference: single-gpu/model.py:365-398)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        self.non_linearity = config.non_linearity.lower()
        if self.non_linearity == "swiglu":
            self.c_fc = nn.Linear(config.n_embd, 2 * config.up_dim, bias=False)
        else:
            self.c_fc = nn.Linear(config.n_embd, config.up_dim, bias=False)
            self._act = ops.get_activation(self.non_linearity)
        self.c_proj = nn.Linear(config.up_dim, config.n_embd, bias=False)
        self.dropout = nn.Dropout(config.dropout)
        self.p_drop = config.dropout

    def forward(self, x, residual=None):
        """``residual`` (the block's residual stream) is added to the second product inside its GEMM; the
        result is then the updated stream, not the feed-forward branch alone."""
        if residual is not None and self.training and self.p_drop > 0.0:
            return residual + self.forward(x)   # dropout sits between the product and the add
        if self.non_linearity == "gelu" and ops.mlp_gelu_supported(x, self.c_fc.weight, self.c_proj.weight):
            # both products on the own GEMM, GELU / GELU' in its epilogues
            return _drop(ops.mlp_gelu(x, self.c_fc.weight, self.c_proj.weight, sole_use=True, residual=residual),
                         self.p_drop, self.training)
        if self.non_linearity == "swiglu" and ops.mlp_swiglu_supported(x, self.c_fc.weight, self.c_proj.weight):
            # gate and up columns paired inside the c_fc GEMM's tiles, SwiGLU and its gradient in the epilogues
            return _drop(ops.mlp_swiglu(x, self.c_fc.weight, self.c_proj.weight, sole_use=True, residual=residual),
                         self.p_drop, self.training)
        h = ops.linear(x, self.c_fc.weight, sole_use=True)
        h = ops.swiglu(h) if self.non_linearity == "swiglu" else self._act(h)
        return _drop(ops.linear(h, self.c_proj.weight, sole_use=True, residual=residual), self.p_drop, self.training)


class Expert(nn.Module):
    """One expert; exists for the ``experts.{j}.expert.*`` key prefix
    (reference: single-gpu/model.py:400-407)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        self.expert = MLP(config)

    def forward(self, x):
        return self.expert(x)


class MoE(nn.Module):
    """DeepSeekMoE layer: shared + routed experts, classic aux loss or aux-loss-free balancing
    (reference: single-gpu/model.py:409-506).

    ``dispatcher`` is a hook for expert parallelism: when set (see ``parallel/ep.py``) the routed
    experts are evaluated through an all-to-all instead of locally."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        self.config = config
        self.n_shared = config.n_shared
        self.n_routed = config.n_exp - config.n_shared
        self.n_act_routed = config.n_act - config.n_shared
        assert self.n_act_routed > 0, "Number of active experts must be greater than shared experts"
        self.experts = nn.ModuleList([Expert(config) for _ in range(config.n_exp)])
        self.gate = nn.Linear(config.n_embd, self.n_routed, bias=False)
        if config.aux_free:
            self.register_buffer("expert_bias", torch.zeros(self.n_routed))
        self.dispatcher = None

    def route(self, x_flat):
        """-> (topk_indices [N,k], topk_gates [N,k], aux_loss)."""
        cfg = self.config
        n_tokens = x_flat.shape[0]
        router_logits = ops.linear(x_flat, self.gate.weight).float()
        E, k = self.n_routed, self.n_act_routed
        # top-k of (logits + selection bias), gates = softmax of the selected raw logits
        topk_indices, topk_gates = ops.topk_gate(router_logits, self.expert_bias if cfg.aux_free else None, k)
        self._layout = None
        if self.dispatcher is None and x_flat.is_cuda:
            # reused by routed_experts_local; the row-block size is the one of the kernels that will run there
            self._layout = (topk_indices, ops.moe_plan(topk_indices, E, self._layout_block(x_flat)))
        with torch.no_grad():
            counts = self._layout[1][5] if self._layout is not None else ops.expert_counts(topk_indices, E)
            fi = counts.float() / n_tokens
            if cfg.aux_free and self.training and not _in_recompute():
                self.expert_bias += (cfg.gamma * (1.0 / E - fi)).to(self.expert_bias.dtype)
        pi = F.softmax(router_logits, dim=1).mean(dim=0)
        aux_loss = (cfg.alpha if cfg.aux_free else cfg.coeff) * E * torch.sum(pi * fi)
        return topk_indices, topk_gates.to(x_flat.dtype), aux_loss

    def _layout_block(self, x_flat) -> int:
        routed = self.experts[self.n_shared:]
        fcs = [e.expert.c_fc.weight for e in routed]
        projs = [e.expert.c_proj.weight for e in routed]
        return ops.grouped_layout_block(x_flat, fcs, projs, routed[0].expert.non_linearity, self.n_act_routed)

    def routed_experts_local(self, x_flat, topk_indices, topk_gates):
        """Routed experts on this rank. GPU path: grouped expert GEMMs over the padded expert-sorted
        layout (no host sync). Fallback (CPU, activations without a
        grouped kernel): sort the (token, slot) pairs by expert once, run every expert on its
        contiguous slice, scale by the gate and scatter-add back, with one host read of the
        per-expert counts instead of the reference's per-expert ``nonzero``."""
        N, k = topk_indices.shape
        routed = self.experts[self.n_shared:]
        fcs = [e.expert.c_fc.weight for e in routed]
        projs = [e.expert.c_proj.weight for e in routed]
        act = routed[0].expert.non_linearity
        if ops.grouped_supported(x_flat, fcs, projs, act):
            # padded expert-sorted layout + grouped MFMA GEMMs: no host sync, no expert loop (dropout on the
            # experts' outputs included: its mask is drawn on the device in the same layout)
            layout = None
            if getattr(self, "_layout", None) is not None and self._layout[0] is topk_indices:
                layout = self._layout[1]
            self._layout = None
            p_drop = self.config.dropout if self.training else 0.0
            return ops.grouped_experts(x_flat, topk_indices, topk_gates, fcs, projs, act, layout, p_drop=p_drop)
        flat_e = topk_indices.reshape(-1)
        order = torch.argsort(flat_e, stable=True)
        tok = torch.div(order, k, rounding_mode="floor")
        counts = ops.expert_counts(flat_e, self.n_routed).tolist()
        xs = x_flat[tok]
        gs = topk_gates.reshape(-1)[order].unsqueeze(1)
        outs = []
        start = 0
        for i, c in enumerate(counts):
            # an expert without tokens still runs (on an empty slice): its parameters stay in the
            # autograd graph and receive zero gradients, so every rank's gradient hooks fire for the
            # same parameters in the same order and the engines issue identical collective sequences
            outs.append(self.experts[i + self.n_shared](xs[start:start + c]))
            start += c
        ys = torch.cat(outs, dim=0) * gs
        return torch.zeros_like(x_flat).index_add_(0, tok, ys.to(x_flat.dtype))

    def forward(self, x):
        B, T, C = x.shape
        x_flat = x.reshape(-1, C)
        topk_indices, topk_gates, aux_loss = self.route(x_flat)
        def shared(rows):
            out = None
            for i in range(self.n_shared):
                o = self.experts[i](rows)
                out = o if out is None else out + o
            return out

        if self.dispatcher is not None:
            # Expert parallel: the dispatch exchange travels under the first half of the shared experts' rows, the
            # return exchange under the second half (and in backward their reverses under the halves' backward).
            N = x_flat.shape[0]
            half = (N // 2 + 255) // 256 * 256 if N >= 1024 else N // 2   # whole GEMM row tiles where it matters
            pending = self.dispatcher.begin(self, x_flat, topk_indices, topk_gates)
            if self.n_shared > 0 and 0 < half < N and self.dispatcher.world > 1:   # (one rank: nothing travels)
                s1 = shared(x_flat[:half])
                pending = self.dispatcher.mid(pending)
                s2 = shared(x_flat[half:])
                shared_output = torch.cat([s1, s2], dim=0)
            else:
                shared_output = shared(x_flat)
                pending = self.dispatcher.mid(pending)
            routed = self.dispatcher.finish(pending)
        else:
            shared_output = shared(x_flat)
            routed = self.routed_experts_local(x_flat, topk_indices, topk_gates)
        y = routed if shared_output is None else shared_output + routed
        return y.view(B, T, C), aux_loss


# ---------------------------------------------------------------------------------------------
class Block(nn.Module):
    """Pre-LN transformer block (reference: single-gpu/model.py:508-533; attention-only
    recomputation as in multi-gpu/ddp/kaggle-ddp.py:526-542).

    The residual stream is passed as ``(x, delta)``: the caller's pending residual update
    ``delta`` is added inside this block's first LayerNorm kernel, and the block returns its own
    pending update instead of adding it."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        self.is_moe = config.moe
        # attention-only recomputation lives here; whole-block recomputation in LLM.forward
        self.act_recomp = config.act_recomp and getattr(config, "act_recomp_scope", "attn") == "attn"
        self.attn = Attention(config)
        self.gqa = config.attn in ("mha", "mqa", "gqa")
        # off while dropout is active in training (it sits between the branch and the add)
        self.fuse_residual = not config.dropout > 0.0
        self.ln1 = nn.LayerNorm(config.n_embd)
        self.ln2 = nn.LayerNorm(config.n_embd)
        if config.moe:
            self.moe = MoE(config)
        else:
            self.mlp = MLP(config)

    # ---- cached decode step on the fused decode kernels ----------------------------------------------
    def _decode_ok(self, x) -> bool:
        """One new token per sequence, at most 64 sequences, dense GQA-family block with a GELU or SwiGLU
        feed-forward, bf16 on the GPU: the step runs as five launches (``ops.decode_linear``)."""
        if not (self.gqa and not self.is_moe and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] == 1
                and x.shape[0] <= 64 and self.mlp.non_linearity in ("gelu", "swiglu")):
            return False
        key = (x.shape[0], x.device)
        ok = getattr(self, "_decode_ok_cache", {}).get(key)
        if ok is None:
            a, m = self.attn.attn, self.mlp
            x2 = x.reshape(x.shape[0], -1)
            act = ops.ACT_SWIGLU if m.non_linearity == "swiglu" else ops.ACT_GELU
            h = x2.new_empty(x2.shape[0], m.c_proj.weight.shape[1])
            ok = (ops.decode_linear_supported(x2, a.c_attn.weight, has_ln=True) and ops.decode_linear_supported(x2, a.c_proj.weight)
                  and ops.decode_linear_supported(x2, m.c_fc.weight, act, True) and ops.decode_linear_supported(h, m.c_proj.weight)
                  and (a.config.n_kv_heads * a.head_size) % 16 == 0)
            self._decode_ok_cache = {**getattr(self, "_decode_ok_cache", {}), key: ok}
        return ok

    def _decode_step(self, x, delta, rope, cache):
        a, m = self.attn.attn, self.mlp
        B, _, C = x.shape
        nh, nkv, hs = a.config.n_head, a.config.n_kv_heads, a.head_size
        x2 = x.reshape(B, C)
        if delta is not None:
            x2 = x2 + delta.reshape(B, C)
        ln1 = (self.ln1.weight, self.ln1.bias, self.ln1.eps)
        # LayerNorm + qkv projection + bias (+ the rotation of the query and key heads: `rope` holds the angle rows of
        # this position); the new key / value rows land in the cache at the device position
        rot = None if rope is None else (rope[0].reshape(-1), rope[1].reshape(-1), (nh + nkv) * hs, hs)
        q = ops.decode_linear(x2, ln1, a.c_attn.weight, a.c_attn.bias, kv=(cache.k, cache.v, cache.pos, nh * hs), rope=rot)
        q = q.view(B, 1, nh, hs)
        y = ops.cached_attention(q, cache.k, cache.v, cache.len_after)
        x2 = ops.decode_linear(y.reshape(B, C), None, a.c_proj.weight, a.c_proj.bias, residual=x2)
        act = ops.ACT_SWIGLU if m.non_linearity == "swiglu" else ops.ACT_GELU
        h = ops.decode_linear(x2, (self.ln2.weight, self.ln2.bias, self.ln2.eps), m.c_fc.weight, m.c_fc.bias, act=act)
        x2 = ops.decode_linear(h, None, m.c_proj.weight, m.c_proj.bias, residual=x2)
        return x2.view(B, 1, C), None, cache, 0.0

    def forward(self, x, delta=None, rope=None, kv_cache=None, VAL_RUN=False):
        if isinstance(kv_cache, _StaticLayer) and self._decode_ok(x):
            return self._decode_step(x, delta, rope, kv_cache)
        # Residual adds ride in the epilogue of the GEMM that produces the branch (GQA output projection,
        # second feed-forward product) where that GEMM exists and no dropout sits in between: the LayerNorm
        # kernels then read the stream once instead of reading stream + branch and writing the sum back.
        # Elsewhere (latent attention, experts, active dropout) the add stays inside the next LayerNorm kernel.
        # (GPU and at least the GEMM kernel's minimum of rows: decode steps keep the LayerNorm form)
        mode = os.environ.get("DPA_FUSE_RESIDUAL", "1")
        on = self.fuse_residual and mode != "0" and (mode == "force" or (x.is_cuda and x.shape[0] * x.shape[1] >= 512))
        fuse_attn = on and self.gqa
        fuse_mlp = on and not self.is_moe
        if delta is not None:
            x, h = ops.add_layer_norm(x, delta, self.ln1.weight, self.ln1.bias, self.ln1.eps)
        elif fuse_attn:
            x, h = ops.layer_norm_keep(x, self.ln1.weight, self.ln1.bias, self.ln1.eps)
        else:
            h = ops.layer_norm(x, self.ln1.weight, self.ln1.bias, self.ln1.eps)
        res = x if fuse_attn else None
        if self.act_recomp and self.training and kv_cache is None:
            attn_out, new_cache = checkpoint(self.attn, h, rope, kv_cache, VAL_RUN, res, use_reentrant=False,
                                             context_fn=_ckpt_contexts)
        else:
            attn_out, new_cache = self.attn(h, rope, kv_cache, VAL_RUN, res)
        if fuse_attn:
            x = attn_out   # the updated stream
            # (the stream goes on to a second reader either way, the feed-forward GEMM's epilogue or the next block's
            # add + LayerNorm: its gradient comes back through this LayerNorm's backward kernel instead of a separate add)
            x, h2 = ops.layer_norm_keep(x, self.ln2.weight, self.ln2.bias, self.ln2.eps)
        else:
            x, h2 = ops.add_layer_norm(x, attn_out, self.ln2.weight, self.ln2.bias, self.ln2.eps)
        if self.is_moe:
            out, aux_loss = self.moe(h2)
        elif fuse_mlp:
            return self.mlp(h2, x), None, new_cache, 0.0
        else:
            out, aux_loss = self.mlp(h2), 0.0
        return x, out, new_cache, aux_loss


# ---------------------------------------------------------------------------------------------
class LLM(nn.Module):
    """The language model (reference: single-gpu/model.py:535-747)."""

    def __init__(self, config: LLMconfig):
        super().__init__()
        self.config = config
        self.head_size = config.n_embd // config.n_head
        self.tkn_emb = nn.Embedding(config.vocab_size, config.n_embd)
        if config.pos_emb == "learn":
            self.pos_emb = nn.Embedding(config.block_size, config.n_embd)
        elif config.pos_emb == "sin":
            pe = torch.zeros(config.block_size, config.n_embd)
            position = torch.arange(0, config.block_size, dtype=torch.float).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, config.n_embd, 2).float() * (-math.log(10000.0) / config.n_embd))
            pe[:, 0::2] = torch.sin(position * div_term)
            pe[:, 1::2] = torch.cos(position * div_term)
            self.register_buffer("pos_emb", pe)
        elif config.pos_emb == "rope":
            d = config.rope_head_dim if config.attn == "mla" else self.head_size
            assert d % 2 == 0, "head dimension must be even"
            cos_t, sin_t = ops.rope_tables(config.block_size, d)
            # complex table kept only so that the state-dict key set equals the reference's
            self.register_buffer("freqs_cis", torch.complex(cos_t, sin_t))
            self.register_buffer("rope_cos", cos_t, persistent=False)
            self.register_buffer("rope_sin", sin_t, persistent=False)
        self.transformer = nn.ModuleDict(dict(
            drop=nn.Dropout(config.dropout),
            h=nn.ModuleList([Block(config) for _ in range(config.n_layer)]),
            ln_f=nn.LayerNorm(config.n_embd)))
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.tkn_emb.weight = self.lm_head.weight  # weight tying
        self.apply(self._init_weights)
        self.VAL_RUN = False
        self.materialize_logits = False   # True: also return [B, T, V] logits when targets are given
        self.loss_chunk_rows = int(os.environ.get("DPA_LOSS_CHUNK", "65536"))
        # the factor the training loop multiplies the loss by before backward (1 / accumulation steps): the fused
        # head loss writes its gradients with it applied (a different incoming gradient is still honoured)
        self.loss_grad_scale = 1.0
        # set by a data-parallel engine built with vocab_parallel=True: (process group,) of the ranks the
        # vocabulary dimension of the head is split over during training
        self.vocab_parallel = None
        # set by a data-parallel engine that reduces the head's share of the tied weight gradient on its
        # own (tied_tail="split"): maps the tied weight to the tensor the head should use in training
        self.head_weight_hook = None
        self.print_act_recomp = config.act_recomp
        self.print_fused_adamw = True
        self._pos_cache = None

    # ---- construction helpers -------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        """``.to(bf16)`` / ``.half()`` must not degrade the rotary tables (fp32 cos/sin, complex64
        ``freqs_cis``): rebuild them on whatever device the weights end up on."""
        super()._apply(fn, recurse)
        if self.config.pos_emb == "rope" and self.rope_cos is not None:
            dev = self.lm_head.weight.device
            if (self.rope_cos.dtype != torch.float32 or not self.freqs_cis.is_complex()) and dev.type != "meta":
                d = self.rope_cos.shape[1] * 2
                cos_t, sin_t = ops.rope_tables(self.config.block_size, d, device=dev)
                self.rope_cos, self.rope_sin = cos_t, sin_t
                self.freqs_cis = torch.complex(cos_t, sin_t)
        return self

    def _init_weights(self, module):
        if isinstance(module, nn.Linear):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)

    def get_num_params(self):
        """(total, active) parameter counts (reference: single-gpu/model.py:588-617)."""
        n_params = sum(p.numel() for p in self.parameters())
        active = self.tkn_emb.weight.numel()
        if self.config.pos_emb == "learn":
            active += self.pos_emb.weight.numel()
        active += sum(p.numel() for p in self.transformer.ln_f.parameters())
        for block in self.transformer.h:
            active += sum(p.numel() for p in block.attn.parameters())
            active += sum(p.numel() for p in block.ln1.parameters())
            active += sum(p.numel() for p in block.ln2.parameters())
            if block.is_moe:
                active += sum(p.numel() for p in block.moe.gate.parameters())
                for i in range(block.moe.n_shared):
                    active += sum(p.numel() for p in block.moe.experts[i].parameters())
                if block.moe.n_routed > 0:
                    per = sum(p.numel() for p in block.moe.experts[block.moe.n_shared].parameters())
                    active += block.moe.n_act_routed * per
            else:
                active += sum(p.numel() for p in block.mlp.parameters())
        return n_params, active

    def param_groups(self, weight_decay: float):
        """Decay every tensor with >= 2 dims, nothing else (reference: single-gpu/model.py:619-629)."""
        params = [p for p in self.parameters() if p.requires_grad]
        return [{"params": [p for p in params if p.dim() >= 2], "weight_decay": weight_decay},
                {"params": [p for p in params if p.dim() < 2], "weight_decay": 0.0}]

    def configure_optimizers(self, weight_decay, learning_rate, device=None):
        """Stock AdamW over the module's own parameters (CPU / un-wrapped use). The engines in
        ``parallel/`` replace this with the flat fused optimizer."""
        groups = self.param_groups(weight_decay)
        dev = str(device) if device is not None else str(next(self.parameters()).device)
        try:
            opt = torch.optim.AdamW(groups, lr=learning_rate, fused=("cuda" in dev))
        except Exception:
            self.print_fused_adamw = False
            opt = torch.optim.AdamW(groups, lr=learning_rate)
        return opt

    # ---- forward --------------------------------------------------------------------------
    def _start_pos(self, kv_caches) -> int:
        if kv_caches is None or kv_caches[0] is None:
            return 0
        c = kv_caches[0]
        if self.config.attn in ("mha", "mqa", "gqa"):
            return c[0].shape[-2]
        return c["c_kv"].shape[1] if self.config.pos_emb == "rope" else c.shape[1]

    def _pos_table(self, dtype, device):
        """Position table in the compute dtype (learned weight, or the sinusoidal buffer cast once)."""
        if self.config.pos_emb == "learn":
            return self.pos_emb.weight
        if self.config.pos_emb == "sin":
            pc = self._pos_cache
            if pc is None or pc.dtype != dtype or pc.device != device:
                pc = self._pos_cache = self.pos_emb.to(device=device, dtype=dtype)
            return pc
        return None

    def forward(self, idx: torch.Tensor, targets=None, kv_caches=None):
        cfg = self.config
        B, T = idx.shape
        wte = self.lm_head.weight
        if isinstance(kv_caches, StaticKVCache):
            # decode step over the preallocated cache: the position is a device scalar, so the position
            # row / rotary rows are gathered on the device and handed to the kernels as 1-row tables
            assert T == 1, "static cache: one token per step"
            static = kv_caches
            ptab = self._pos_table(wte.dtype, wte.device)
            x = ops.embedding(idx, wte, None if ptab is None else ptab.index_select(0, static.pos), 0)
            rope = None
            if cfg.pos_emb == "rope":
                rope = (self.rope_cos.index_select(0, static.pos), self.rope_sin.index_select(0, static.pos), 0)
            kv_caches = [static.layer(i) for i in range(cfg.n_layer)]
        else:
            start_pos = self._start_pos(kv_caches)
            x = ops.embedding(idx, wte, self._pos_table(wte.dtype, wte.device), start_pos)
            rope = (self.rope_cos, self.rope_sin, start_pos) if cfg.pos_emb == "rope" else None
        x = _drop(x, cfg.dropout, self.training)
        if kv_caches is None:
            kv_caches = [None] * cfg.n_layer
        new_caches = []
        total_aux = 0.0
        delta = None
        # Whole-block recomputation (reference: single-gpu/model.py:677-680): only the block inputs are
        # kept, the block runs again in backward (non-reentrant, so hooks and direct gradient
        # accumulation behave as in a plain backward).
        block_ckpt = (cfg.act_recomp and getattr(cfg, "act_recomp_scope", "attn") == "block" and self.training
                      and torch.is_grad_enabled() and all(c is None for c in kv_caches))
        for i, block in enumerate(self.transformer.h):
            if block_ckpt:
                x, delta, cache, aux = checkpoint(block, x, delta, rope, None, self.VAL_RUN, use_reentrant=False,
                                                  context_fn=_ckpt_contexts)
            else:
                x, delta, cache, aux = block(x, delta, rope, kv_caches[i], self.VAL_RUN)
            new_caches.append(cache)
            total_aux = total_aux + aux
        ln_f = self.transformer.ln_f
        if (targets is None and T == 1 and new_caches and isinstance(new_caches[0], _StaticLayer) and x.dtype == torch.bfloat16
                and os.environ.get("DPA_FUSED_DECODE_HEAD", "0") == "1" and ops.decode_linear_supported(x.reshape(B, -1), wte, has_ln=True)):
            # static-cache decode step: final LayerNorm + head product in one launch (opt-in: at the vocabulary's
            # width the library product runs at HBM speed and the two launches are faster, 18.5 vs 36 us at batch 8)
            x2 = x.reshape(B, -1) if delta is None else x.reshape(B, -1) + delta.reshape(B, -1)
            logits = ops.decode_linear(x2, (ln_f.weight, ln_f.bias, ln_f.eps), wte)
            return logits.view(B, 1, -1), None, new_caches
        if delta is None:
            x = ops.layer_norm(x, ln_f.weight, ln_f.bias, ln_f.eps)
        else:
            _, x = ops.add_layer_norm(x, delta, ln_f.weight, ln_f.bias, ln_f.eps)
        if targets is not None:
            logits = None
            if self.head_weight_hook is not None and self.training and torch.is_grad_enabled():
                wte = self.head_weight_hook(wte)
            if self.materialize_logits:
                logits = ops.linear(x, wte)
                main_loss = ops.cross_entropy(logits, targets)
            elif self.vocab_parallel is not None and self.training and torch.is_grad_enabled():
                main_loss = ops.vocab_parallel_linear_cross_entropy(x, wte, targets, group=self.vocab_parallel[0],
                                                                    chunk_rows=self.loss_chunk_rows)
            else:
                main_loss = ops.linear_cross_entropy(x, wte, targets, chunk_rows=self.loss_chunk_rows,
                                                     expected_grad=self.loss_grad_scale if self.training else 1.0)
            loss = main_loss + total_aux / cfg.n_layer
        else:
            logits = ops.linear(x[:, -1:, :], wte)   # a slice, not a list index: no host-built index tensor
            loss = None
        return logits, loss, new_caches

    # ---- generation -----------------------------------------------------------------------
    @staticmethod
    def _sample(logits: torch.Tensor, temperature: float, topk: int | None, fused: bool = False) -> torch.Tensor:
        """Next token ``[B, 1]`` from the last position's logits (temperature, top-k, multinomial); no
        data-dependent host work, so it can be captured in a hipGraph. ``fused``: the one-kernel sampler
        (same distribution, its own random stream) instead of the chain of stock ops."""
        last = logits[:, -1, :]
        return ops.sample_fused(last, temperature, topk) if fused else ops.sample_eager(last, temperature, topk)

    @torch.no_grad()
    def generate(self, idx: torch.Tensor, max_new_tokens: int, temperature: float = 1.0, topk: int | None = None,
                 hip_graph: bool | None = None):
        """KV-cache sampling with a sliding window of ``block_size - 1`` cached positions
        (reference: single-gpu/model.py:699-747).

        ``hip_graph`` (default: automatic) decodes from ONE captured hipGraph per token over a
        preallocated cache when the model is on the GPU, its key rows fit the flash kernels (head size
        64 / 128, or an MLA latent key of at most 128 columns) and prompt plus new tokens fit the context: a decode step is a few hundred small kernels, bound by launch cost."""
        was_training = self.training
        self.eval()
        try:
            cfg = self.config
            fits = idx.size(1) + max_new_tokens - 1 <= cfg.block_size and max_new_tokens >= 3
            able = idx.is_cuda and StaticKVCache.static_width(cfg) is not None and fits
            if hip_graph is None:
                hip_graph = able
            self.last_decode_graphed = False      # introspection: did the last call replay a graph?
            if hip_graph and able:
                out = self._generate_graphed(idx, max_new_tokens, temperature, topk)
                if out is not None:
                    self.last_decode_graphed = True
                    return out
            return self._generate_eager(idx, max_new_tokens, temperature, topk)
        finally:
            if was_training:
                self.train()

    def _generate_graphed(self, idx, max_new_tokens, temperature, topk):
        """Prompt through the normal forward, then one eager decode step over the static cache (lazy
        initialisations happen outside the capture), then the same step captured once and replayed for
        every remaining token. Returns None if the step cannot be captured."""
        cfg = self.config
        B, P = idx.shape
        dev = idx.device
        logits, _, caches = self.forward(idx)
        static = StaticKVCache(cfg, B, cfg.block_size, dev, self.lm_head.weight.dtype)
        static.fill(caches, P)
        del caches
        out = torch.empty(B, max_new_tokens, dtype=torch.int64, device=dev)
        tok = self._sample(logits, temperature, topk)
        out[:, :1] = tok
        out_pos = torch.ones(1, dtype=torch.int64, device=dev)

        # one seed for the generation (the sampler mixes the device-side position into its stream) and the
        # zeroed counter its last workgroup uses to advance the step's device counters
        seed = torch.randint(-(1 << 62), 1 << 62, (1,), dtype=torch.int64, device=dev)
        ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        one_launch = os.environ.get("DPA_FUSED_DECODE", "1") != "0" and ops.sample_step_supported(logits[:, -1, :])

        def step():
            lg, _, _ = self.forward(tok, kv_caches=static)
            if one_launch:
                # sampling, history write, next input and the counter updates in the sampler's launch
                ops.sample_step(lg[:, -1, :], temperature, topk, seed, tok, out, out_pos, static.pos, static.len_after, ticket)
                return
            nxt = self._sample(lg, temperature, topk, fused=True)
            out.index_copy_(1, out_pos, nxt)
            tok.copy_(nxt)
            static.advance()
            out_pos.add_(1)

        step()                                   # token 2, eagerly
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(graph):
                step()
        except Exception as exc:                 # some op of this configuration cannot be captured
            torch.cuda.synchronize()
            print(f"[hip_graph] decode capture failed ({type(exc).__name__}: {exc}); decoding eagerly", flush=True)
            return None
        for _ in range(max_new_tokens - 2):
            graph.replay()
        return torch.cat((idx, out), dim=1)

    def _generate_eager(self, idx, max_new_tokens, temperature, topk):
        cfg = self.config
        kv_caches = [None] * cfg.n_layer
        for i in range(max_new_tokens):
            inp = (idx if idx.size(1) <= cfg.block_size else idx[:, -cfg.block_size:]) if i == 0 else idx[:, -1:]
            if kv_caches[0] is not None and self._start_pos(kv_caches) >= cfg.block_size:
                keep = cfg.block_size - 1
                for li in range(cfg.n_layer):
                    c = kv_caches[li]
                    if cfg.attn in ("mha", "mqa", "gqa"):
                        kv_caches[li] = (c[0][..., -keep:, :], c[1][..., -keep:, :])
                    elif cfg.pos_emb == "rope":
                        kv_caches[li] = {"c_kv": c["c_kv"][:, -keep:, :], "k_r": c["k_r"][:, :, -keep:, :]}
                    else:
                        kv_caches[li] = c[:, -keep:, :]
            logits, _, kv_caches = self.forward(inp, kv_caches=kv_caches)
            idx = torch.cat((idx, self._sample(logits, temperature, topk)), dim=1)
        return idx
