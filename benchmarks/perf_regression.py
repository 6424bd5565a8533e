#!/usr/bin/env python3
"""Performance regression harness: a fixed set of kernel micro-benchmarks plus two short end-to-end runs on one
MI355X, compared against a committed table.

    python benchmarks/perf_regression.py                 # measure, compare with benchmarks/perf_baseline_mi355x.json
    python benchmarks/perf_regression.py --update        # measure and rewrite the table
    python benchmarks/perf_regression.py --only gemm     # a subset (substring of the entry names; with --update the other entries are kept)
    python benchmarks/perf_regression.py --tolerance 0.1 # allowed slowdown (default 0.08: boxes of one pool differ by 2-4 %)

Every entry is a time in microseconds (median of a few timed calls; decode-step kernels are replayed from a hipGraph
of 12 calls because their launch path is what matters). Exit status 1 if any entry is slower than the table allows;
entries missing from the table are reported and do not fail. The two end-to-end entries run `bench.py` in a child
process (GPT-2-124M at micro-batch 64 and the graphed decode), everything else runs in this process.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from distributed_pytorch_amd import ops  # noqa: E402
from distributed_pytorch_amd._ext import ops as hip  # noqa: E402

TABLE = os.path.join(ROOT, "benchmarks", "perf_baseline_mi355x.json")
DEV = "cuda"


def median_us(fn, n=15, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def graph_us(fn, reps=12, iters=40):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters * reps)


def bf(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).bfloat16()


# ---- the entries: name -> callable returning microseconds --------------------------------------------------------
def entries():
    E = {}
    M = 65536

    def gemm(name, N, K, epi, bias, aux):
        def run():
            a, w = bf(M, K), bf(N, K, scale=K ** -0.5)
            b = bf(N) if bias else None
            x = bf(M, N) if aux else None
            return median_us(lambda: hip().gemm_nt(a, w, b, epi, x))
        E[f"gemm_nt {name} M{M} N{N} K{K}"] = run

    gemm("qkv +bias", 2304, 768, 0, True, False)
    gemm("attention projection +bias +residual", 768, 768, 3, True, True)
    gemm("c_fc GELU", 3072, 768, 1, False, False)
    gemm("c_proj +residual", 768, 3072, 3, False, True)
    gemm("dGELU (c_proj input gradient)", 3072, 768, 2, False, True)

    def logits():
        a, w = bf(16384, 768), bf(50304, 768, scale=0.03)
        return median_us(lambda: hip().gemm_nt(a, w, None, 0, None), n=8)
    E["gemm_nt logits M16384 N50304 K768"] = logits

    def rope_gemm():
        from distributed_pytorch_amd.ops.rope import rope_tables
        a, w, b = bf(M, 768), bf(1280, 768, scale=0.03), bf(1280)
        c, s = rope_tables(1024, 64, device=DEV)
        return median_us(lambda: hip().gemm_nt_rope(a, w, b, c, s, 0, 1024, 1024, 64))
    E["gemm_nt qkv +bias +rotary (GQA 12/4) M65536 N1280 K768"] = rope_gemm

    def wgrad(N, K):
        def run():
            dy, x = bf(M, N), bf(M, K)
            return median_us(lambda: hip().wgrad2(dy, x), n=10)
        E[f"weight gradient M{M} N{N} K{K}"] = run

    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        wgrad(N, K)

    def grouped_wgrad():
        seg = torch.full((8,), 16384, dtype=torch.int64)
        start = torch.cumsum(seg, 0) - seg
        dy, x = bf(131072, 3072), bf(131072, 768)
        ss, sr = start.to(DEV), seg.to(DEV)
        return median_us(lambda: hip().grouped_wgrad(dy, x, ss, sr), n=10)
    E["grouped weight gradient 8 x 16384 rows N3072 K768"] = grouped_wgrad

    def attn(B, T, nh, nkv, hs):
        def setup():
            q, k, v = bf(B, T, nh, hs), bf(B, T, nkv, hs), bf(B, T, nkv, hs)
            sc = 1 / math.sqrt(hs)
            o, lse = hip().attn_fwd(q, k, v, sc)
            return q, k, v, sc, o, lse

        def fwd():
            q, k, v, sc, o, lse = setup()
            return median_us(lambda: hip().attn_fwd(q, k, v, sc))

        def bwd():
            q, k, v, sc, o, lse = setup()
            do = torch.randn_like(o)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            return median_us(lambda: hip().attn_bwd(do, q, k, v, o, lse, dq, dk, dv, sc))
        tag = f"B{B} T{T} nh{nh} nkv{nkv} hs{hs}"
        E[f"attention forward {tag}"] = fwd
        E[f"attention backward {tag}"] = bwd

    attn(64, 1024, 12, 12, 64)
    attn(16, 2048, 32, 8, 128)

    def ln():
        x = bf(M, 768).requires_grad_()
        w, b = bf(768).requires_grad_(), bf(768).requires_grad_()
        dy = bf(M, 768)

        def step():
            x.grad = w.grad = b.grad = None
            ops.layer_norm(x, w, b).backward(dy)
        return median_us(step)
    E["LayerNorm forward + backward M65536 C768"] = ln

    def xent():
        lg = bf(16384, 50304)
        t = torch.randint(0, 50304, (16384,), device=DEV)

        def step():
            l2 = lg.detach().requires_grad_()
            ops.cross_entropy(l2, t).backward()
        return median_us(step, n=8)
    E["cross-entropy forward + backward 16384 x 50304"] = xent

    def adamw():
        from distributed_pytorch_amd.parallel.engine import Engine
        from distributed_pytorch_amd.config import PRESETS, LLMconfig
        from distributed_pytorch_amd.models.llm import LLM
        cfg = LLMconfig(**{**LLMconfig().to_dict(), **PRESETS["gpt2-124m"]}).finalize()
        eng = Engine(LLM(cfg), dtype="bf16", device="cuda:0")
        opt = eng.configure_optimizers(learning_rate=1e-4)
        for g in eng.groups:
            g.rgrad.normal_(std=1e-3)

        def step():
            eng.clip_grad_norm_(1.0)
            opt.step()
        return median_us(step, n=8)
    E["gradient norm + clip + AdamW, 124M parameters"] = adamw

    def dec(name, K, N, ln_, bias, res, act):
        def run():
            x, w = bf(8, K), bf(N * (2 if act == ops.ACT_SWIGLU else 1), K, scale=K ** -0.5)
            lnp = (bf(K), bf(K), 1e-5) if ln_ else None
            b = bf(w.shape[0]) if bias else None
            r = bf(8, N) if res else None
            return graph_us(lambda: ops.decode_linear(x, lnp, w, b, r, act))
        E[f"decode step (graph) {name} B8 K{K} N{N}"] = run

    def dec64():
        x, w, b = bf(64, 768), bf(2304, 768, scale=768 ** -0.5), bf(2304)
        lnp = (bf(768), bf(768), 1e-5)
        return graph_us(lambda: ops.decode_linear(x, lnp, w, b, None, ops.ACT_NONE))
    E["decode step (graph) LayerNorm + qkv + bias B64 K768 N2304"] = dec64

    dec("LayerNorm + qkv + bias", 768, 2304, True, True, False, ops.ACT_NONE)
    dec("projection + bias + residual", 768, 768, False, True, True, ops.ACT_NONE)
    dec("LayerNorm + c_fc + GELU", 768, 3072, True, False, False, ops.ACT_GELU)
    dec("c_proj + residual", 3072, 768, False, False, True, ops.ACT_NONE)

    def dec_attn():
        q, kc, vc = bf(8, 1, 12, 64), bf(8, 1024, 12, 64), bf(8, 1024, 12, 64)
        n = torch.tensor([400], dtype=torch.int32, device=DEV)
        return graph_us(lambda: ops.cached_attention(q, kc, vc, n))
    E["decode step (graph) attention B8 nh12 hs64, 400 cached keys"] = dec_attn

    def sampler():
        lg = bf(8, 50304, scale=2.0)
        seed = torch.tensor([1], dtype=torch.int64, device=DEV)
        return graph_us(lambda: ops.sample_fused(lg, 0.8, 50, seed))
    E["decode step (graph) sampler B8 V50304 top-k 50"] = sampler

    def child(args, key):
        def run():
            r = subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            return key(r.stdout)
        return run

    E["training step GPT-2-124M micro-batch 64 (bench.py, median)"] = child(
        ["bench.py", "--batch", "64", "--steps", "20", "--warmup", "5"],
        lambda out: json.loads(out.strip().splitlines()[-1])["step_ms"]["median"] * 1e3)

    def gen_key(out):
        line = [ln_ for ln_ in out.splitlines() if "hipGraph" in ln_][-1]
        return float(line.split("=")[1].split("ms/step")[0]) * 1e3
    E["generation GPT-2-124M B8, 256 tokens, per step (generate_bench.py, graphed)"] = child(
        ["benchmarks/generate_bench.py", "gpt2-124m", "8", "128", "256"], gen_key)
    return E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--update", action="store_true", help="rewrite the table with this run's times")
    ap.add_argument("--only", type=str, default="", help="run the entries whose name contains this")
    ap.add_argument("--tolerance", type=float, default=0.08)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "the harness measures on the GPU"
    table = {}
    if os.path.exists(TABLE):
        table = json.load(open(TABLE))["entries"]
    got, failed = {}, []
    print(f"{'entry':84s} {'now us':>10s} {'table us':>10s} {'ratio':>7s}")
    for name, fn in entries().items():
        if args.only and args.only not in name:
            continue
        try:
            torch.manual_seed(0)
            us = float(fn())
        except Exception as exc:                      # an entry that cannot run is a failure, not a skip
            print(f"{name:84s} ERROR {type(exc).__name__}: {str(exc)[:80]}")
            failed.append(name)
            continue
        finally:
            torch.cuda.empty_cache()
        got[name] = us
        ref = table.get(name)
        ratio = us / ref if ref else float("nan")
        mark = ""
        if ref and ratio > 1 + args.tolerance:
            mark = "  SLOWER"
            failed.append(name)
        elif ref and ratio < 1 - args.tolerance:
            mark = "  faster"
        print(f"{name:84s} {us:10.1f} {ref if ref else float('nan'):10.1f} {ratio:7.3f}{mark}", flush=True)
    if args.update:
        merged = {**table, **got}
        json.dump({"device": torch.cuda.get_device_name(0), "unit": "us", "entries": merged}, open(TABLE, "w"), indent=1,
                  sort_keys=True)
        print(f"table written: {TABLE} ({len(merged)} entries)")
        return 0
    if failed:
        print(f"{len(failed)} entr{'y' if len(failed) == 1 else 'ies'} outside the tolerance of {args.tolerance:.0%}: " + "; ".join(failed))
        return 1
    print("all entries within tolerance")
    return 0


if __name__ == "__main__":
    sys.exit(main())
