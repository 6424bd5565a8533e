"""Numerics of every HIP kernel against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from distributed_pytorch_amd import _ext
    return _ext.require()


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,C", [(1024, 768), (37, 256), (4096, 1024), (513, 2048), (64, 4096)])
def test_layernorm_fwd_bwd(dtype, rows, C):
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    x = torch.randn(rows, C, device=DEV, dtype=dtype) * 2 + 0.5
    w = (torch.randn(C, device=DEV) * 0.5 + 1).to(dtype)
    b = (torch.randn(C, device=DEV) * 0.1).to(dtype)
    dy = torch.randn(rows, C, device=DEV, dtype=dtype)
    xr, wr, br = [t.float().detach().requires_grad_() for t in (x, w, b)]
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    yr.backward(dy.float())
    xs, ws, bs = [t.detach().clone().requires_grad_() for t in (x, w, b)]
    y = ops.layer_norm(xs, ws, bs, 1e-5)
    y.backward(dy)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel_err(y, yr) < tol
    assert rel_err(xs.grad, xr.grad) < tol
    assert rel_err(ws.grad, wr.grad) < tol
    assert rel_err(bs.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_add_layernorm(dtype):
    from distributed_pytorch_amd import ops
    torch.manual_seed(1)
    rows, C = 777, 768
    x = torch.randn(rows, C, device=DEV, dtype=dtype)
    r = torch.randn(rows, C, device=DEV, dtype=dtype)
    w = (torch.rand(C, device=DEV) + 0.5).to(dtype)
    b = torch.randn(C, device=DEV).to(dtype)
    g1 = torch.randn(rows, C, device=DEV, dtype=dtype)
    g2 = torch.randn(rows, C, device=DEV, dtype=dtype)
    xr, rr, wr, br = [t.float().detach().requires_grad_() for t in (x, r, w, b)]
    sr = (xr + rr).to(dtype).float() if dtype == torch.bfloat16 else xr + rr
    sr = xr + rr
    yr = F.layer_norm(sr, (C,), wr, br, 1e-5)
    (sr * g1.float() + yr * g2.float()).sum().backward()
    xs, rs, ws, bs = [t.detach().clone().requires_grad_() for t in (x, r, w, b)]
    s, y = ops.add_layer_norm(xs, rs, ws, bs, 1e-5)
    (s.float() * g1.float() + y.float() * g2.float()).sum().backward()
    tol = 1.5e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel_err(s, sr) < tol and rel_err(y, yr) < tol
    assert rel_err(xs.grad, xr.grad) < tol and rel_err(rs.grad, rr.grad) < tol
    assert rel_err(ws.grad, wr.grad) < tol and rel_err(bs.grad, br.grad) < tol


@pytest.mark.parametrize("gdtype", [torch.bfloat16, torch.float32])
def test_fused_adamw_matches_torch(gdtype):
    from distributed_pytorch_amd import ops
    torch.manual_seed(2)
    n, d = 8 * 1000, 8 * 600
    p0 = torch.randn(n, device=DEV)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([{"params": [ref], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    # two segments with different decay: emulate with two torch params
    ref_a = p0[:d].clone().requires_grad_()
    ref_b = p0[d:].clone().requires_grad_()
    opt = torch.optim.AdamW([{"params": [ref_a], "weight_decay": 0.1}, {"params": [ref_b], "weight_decay": 0.0}],
                            lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    master = p0.clone()
    m = torch.zeros_like(master)
    v = torch.zeros_like(master)
    pout = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    for step in range(1, 6):
        g = (torch.randn(n, device=DEV) * 0.1).to(gdtype)
        ref_a.grad = g[:d].float().clone()
        ref_b.grad = g[d:].float().clone()
        opt.step()
        ops.adamw_step(master, m, v, g, pout, [(d, 0.1), (n, 0.0)], 1e-2, 0.9, 0.95, 1e-8, step)
    want = torch.cat([ref_a.detach(), ref_b.detach()])
    assert rel_err(master, want) < 1e-5
    assert rel_err(pout, want) < 5e-3


def test_adamw_clip_and_overflow_skip():
    from distributed_pytorch_amd import ops
    n = 4096
    master = torch.ones(n, device=DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    g = torch.full((n,), 2.0, device=DEV)
    norm2 = torch.zeros(1, device=DEV)
    ops.sumsq(g, norm2)
    assert abs(norm2.item() - 4.0 * n) < 1e-2 * n
    # clip to norm 1: effective grad = g / ||g||
    ops.adamw_step(master, m, v, g, None, [(n, 0.0)], 0.1, 0.9, 0.999, 1e-8, 1, norm2, 1.0, 1.0)
    geff = 2.0 / math.sqrt(4.0 * n)
    assert torch.allclose(m, torch.full_like(m, 0.1 * geff), rtol=1e-4)
    before = master.clone()
    bad = torch.tensor([float("inf")], device=DEV)
    ops.adamw_step(master, m, v, g, None, [(n, 0.0)], 0.1, 0.9, 0.999, 1e-8, 2, bad, 1.0, 1.0)
    assert torch.equal(master, before)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,V", [(64, 50304), (300, 4096), (17, 512)])
def test_cross_entropy(dtype, rows, V):
    from distributed_pytorch_amd import ops
    torch.manual_seed(3)
    logits = (torch.randn(rows, V, device=DEV) * 3).to(dtype)
    tgt = torch.randint(0, V, (rows,), device=DEV)
    tgt[::5] = -1
    lr_ = logits.float().detach().requires_grad_()
    want = F.cross_entropy(lr_, tgt, ignore_index=-1)
    want.backward()
    ls = logits.detach().clone().requires_grad_()
    got = ops.cross_entropy(ls, tgt)
    got.backward()
    assert abs(got.item() - want.item()) < 2e-3 * max(1.0, abs(want.item()))
    tol = 1.5e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(ls.grad, lr_.grad) < tol
    assert torch.equal(ls.detach(), logits)  # input logits untouched


def _unfused_head_loss(x, w, tgt, scale, rows=8192):
    """fp32 lm_head + cross-entropy and its gradients, computed in row chunks (the full logits would not fit)."""
    xr = x.detach().float()
    wr = w.detach().float()
    n_valid = (tgt != -1).sum().clamp_min(1).float()
    loss = torch.zeros((), device=x.device)
    dx = torch.empty_like(xr)
    dw = torch.zeros_like(wr)
    for r0 in range(0, xr.shape[0], rows):
        xc = xr[r0:r0 + rows].requires_grad_()
        wc = wr.clone().requires_grad_()
        part = F.cross_entropy(xc @ wc.t(), tgt[r0:r0 + rows], ignore_index=-1, reduction="sum") / n_valid
        (part * scale).backward()
        loss += part.detach()
        dx[r0:r0 + rows] = xc.grad
        dw += wc.grad
    return loss, dx, dw


@pytest.mark.parametrize("N,C,chunk,expected", [
    (3000, 256, 1024, 1.0),          # ragged chunks of a small problem
    (3000, 256, 1024, 0.5),          # gradient factor announced in advance (no rescaling pass)
    (4096 + 200, 768, 2048, 0.25),   # whole chunks on the 256 x 256 weight-gradient kernel plus a ragged one
    (65536 + 3000, 768, 65536, 1.0),  # the headline's chunk size: two chunks, the second ragged
])
def test_linear_cross_entropy_matches_unfused(N, C, chunk, expected):
    from distributed_pytorch_amd import ops
    torch.manual_seed(4)
    V = 50304
    x = (torch.randn(N, C, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(V, C, device=DEV) * 0.05).to(torch.bfloat16).requires_grad_()
    tgt = torch.randint(0, V, (N,), device=DEV)
    tgt[:7] = -1
    loss = ops.linear_cross_entropy(x, w, tgt, chunk_rows=chunk, expected_grad=expected)
    (loss * 0.5).backward()          # not always the announced factor: the backward must honour what arrives
    want, dxr, dwr = _unfused_head_loss(x, w, tgt, 0.5)
    assert abs(loss.item() - want.item()) < 5e-3 * want.item()
    assert rel_err(x.grad, dxr) < 2e-2
    assert rel_err(w.grad, dwr) < 2e-2


def test_linear_cross_entropy_backward_twice_over_a_retained_graph():
    """The stored gradients are rescaled from the factor they carry, so a second backward with another incoming
    gradient gives that gradient's result (accumulated on top of the first)."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(5)
    N, C, V = 2048, 256, 1024
    x = (torch.randn(N, C, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(V, C, device=DEV) * 0.05).to(torch.bfloat16).requires_grad_()
    tgt = torch.randint(0, V, (N,), device=DEV)
    loss = ops.linear_cross_entropy(x, w, tgt, expected_grad=0.5)
    (loss * 0.5).backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    (loss * 2.0).backward()
    assert rel_err(x.grad, g1.float() * 4.0) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_embedding_fwd_bwd(dtype):
    from distributed_pytorch_amd import ops
    torch.manual_seed(5)
    V, C, B, T = 1000, 768, 4, 128
    wte = torch.randn(V, C, device=DEV).to(dtype).requires_grad_()
    wpe = torch.randn(256, C, device=DEV).to(dtype).requires_grad_()
    idx = torch.randint(0, 50, (B, T), device=DEV)   # many repeats -> long runs in the sorted backward
    out = ops.embedding(idx, wte, wpe, 3)
    g = torch.randn_like(out)
    out.backward(g)
    wr = wte.detach().float().requires_grad_()
    pr = wpe.detach().float().requires_grad_()
    want = F.embedding(idx, wr) + pr[3:3 + T]
    want.backward(g.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(out, want) < tol
    assert rel_err(wte.grad, wr.grad) < tol
    assert rel_err(wpe.grad, pr.grad) < tol


@pytest.mark.parametrize("N,hot", [(1000, 1.0), (4099, 0.6), (33, 0.0), (64, 1.0)])
def test_embedding_bwd_skewed_runs(N, hot):
    """Long runs of one token crossing many chunks of the sorted backward, ragged N."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(55)
    V, C = 300, 256
    idx = torch.randint(0, V, (1, N), device=DEV)
    idx[0, torch.rand(N, device=DEV) < hot] = 7
    wte = torch.randn(V, C, device=DEV).requires_grad_()
    out = ops.embedding(idx, wte, None, 0)
    g = torch.randn_like(out)
    out.backward(g)
    wr = wte.detach().clone().requires_grad_()
    F.embedding(idx, wr).backward(g)
    assert rel_err(wte.grad, wr.grad) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gelu_swiglu(dtype):
    from distributed_pytorch_amd import ops
    torch.manual_seed(6)
    x = (torch.randn(333, 1024, device=DEV) * 2).to(dtype).requires_grad_()
    g = torch.randn(333, 1024, device=DEV).to(dtype)
    y = ops.gelu(x)
    y.backward(g)
    xr = x.detach().float().requires_grad_()
    yr = F.gelu(xr)
    yr.backward(g.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(y, yr) < tol and rel_err(x.grad, xr.grad) < tol
    h = (torch.randn(200, 2 * 384, device=DEV)).to(dtype).requires_grad_()
    g2 = torch.randn(200, 384, device=DEV).to(dtype)
    z = ops.swiglu(h)
    z.backward(g2)
    hr = h.detach().float().requires_grad_()
    a, b = hr.chunk(2, dim=-1)
    zr = F.silu(a) * b
    zr.backward(g2.float())
    assert rel_err(z, zr) < tol and rel_err(h.grad, hr.grad) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rope_matches_complex_rotation(dtype):
    from distributed_pytorch_amd import ops
    torch.manual_seed(7)
    B, T, H, hs, pos0 = 2, 50, 6, 64, 5
    x = torch.randn(B, T, H, hs, device=DEV).to(dtype).requires_grad_()
    cos_t, sin_t = ops.rope_tables(128, hs, device=DEV)
    y = ops.apply_rope(x, cos_t, sin_t, pos0)
    g = torch.randn_like(y)
    y.backward(g)
    # reference: complex multiplication as in the reference model
    freqs = torch.polar(torch.ones_like(cos_t), torch.atan2(sin_t, cos_t))[pos0:pos0 + T]
    xr = x.detach().float().requires_grad_()
    xc = torch.view_as_complex(xr.reshape(B, T, H, hs // 2, 2))
    yr = torch.view_as_real(xc * freqs.view(1, T, 1, hs // 2)).flatten(3)
    yr.backward(g.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(y, yr) < tol and rel_err(x.grad, xr.grad) < tol


def _attn_ref(q, k, v, scale):
    B, Tq, nh, hs = q.shape
    Tk, nkv = k.shape[1], k.shape[2]
    qh = q.float().transpose(1, 2)
    kh = k.float().transpose(1, 2).repeat_interleave(nh // nkv, dim=1)
    vh = v.float().transpose(1, 2).repeat_interleave(nh // nkv, dim=1)
    s = (qh @ kh.transpose(-1, -2)) * scale
    mask = torch.ones(Tq, Tk, dtype=torch.bool, device=q.device).tril(diagonal=Tk - Tq)
    s = s.masked_fill(~mask, float("-inf"))
    return (torch.softmax(s, dim=-1) @ vh).transpose(1, 2)


@pytest.mark.parametrize("B,Tq,Tk,nh,nkv,hs", [
    (2, 256, 256, 4, 4, 64),
    (1, 1024, 1024, 12, 12, 64),
    (2, 200, 200, 8, 2, 64),      # GQA + ragged T
    (2, 384, 384, 4, 1, 128),     # MQA, hs 128
    (1, 130, 130, 2, 2, 128),
    (2, 1, 77, 4, 2, 64),         # single-token decode against a cache
    (1, 40, 300, 4, 4, 64),       # chunked prefill against a cache
])
def test_flash_attention_fwd_bwd(B, Tq, Tk, nh, nkv, hs):
    from distributed_pytorch_amd import ops
    torch.manual_seed(8)
    q = torch.randn(B, Tq, nh, hs, device=DEV).to(torch.bfloat16).requires_grad_()
    k = torch.randn(B, Tk, nkv, hs, device=DEV).to(torch.bfloat16).requires_grad_()
    v = torch.randn(B, Tk, nkv, hs, device=DEV).to(torch.bfloat16).requires_grad_()
    scale = 1.0 / math.sqrt(hs)
    o = ops.causal_attention(q, k, v, scale)
    g = torch.randn_like(o)
    o.backward(g)
    qr, kr, vr = [t.detach().float().requires_grad_() for t in (q, k, v)]
    orf = _attn_ref(qr, kr, vr, scale)
    orf.backward(g.float())
    assert rel_err(o, orf) < 1e-2
    assert rel_err(q.grad, qr.grad) < 2e-2
    assert rel_err(k.grad, kr.grad) < 2e-2
    assert rel_err(v.grad, vr.grad) < 2e-2


@pytest.mark.parametrize("B,Tq,Tk,nh,nkv,dk,dv,p_drop", [
    (2, 384, 384, 4, 4, 256, 128, 0.0),    # latent attention with 128 rotary columns behind 128-wide heads
    (1, 300, 300, 8, 8, 256, 128, 0.0),    # ragged T
    (2, 256, 256, 4, 2, 256, 128, 0.0),    # grouped keys
    (2, 320, 320, 4, 4, 128, 64, 0.0),
    (1, 64, 200, 4, 4, 256, 128, 0.0),     # queries at the end of a longer key range
    (2, 200, 200, 4, 4, 192, 100, 0.0),    # zero-padded up to (256, 128)
])
def test_flash_attention_key_wider_than_value(B, Tq, Tk, nh, nkv, dk, dv, p_drop):
    from distributed_pytorch_amd import ops
    torch.manual_seed(18)
    q = torch.randn(B, Tq, nh, dk, device=DEV).to(torch.bfloat16).requires_grad_()
    k = torch.randn(B, Tk, nkv, dk, device=DEV).to(torch.bfloat16).requires_grad_()
    v = torch.randn(B, Tk, nkv, dv, device=DEV).to(torch.bfloat16).requires_grad_()
    scale = 1.0 / math.sqrt(dk)
    o = ops.causal_attention(q, k, v, scale, p_drop)
    assert o.shape == (B, Tq, nh, dv)
    g = torch.randn_like(o)
    o.backward(g)
    qr, kr, vr = [t.detach().float().requires_grad_() for t in (q, k, v)]
    orf = _attn_ref(qr, kr, vr, scale)
    orf.backward(g.float())
    assert rel_err(o, orf) < 1e-2
    assert rel_err(q.grad, qr.grad) < 2e-2
    assert rel_err(k.grad, kr.grad) < 2e-2
    assert rel_err(v.grad, vr.grad) < 2e-2


def test_flash_attention_key_wider_than_value_dropout_is_consistent():
    """Dropout on the (256, 128) kernels: forward and both backward kernels regenerate the same mask (the result
    equals the explicit-mask reference built from the mask the kernels report for the seed)."""
    from distributed_pytorch_amd import ops
    from distributed_pytorch_amd.ops.attention import dropout_keep_mask
    torch.manual_seed(19)
    B, T, nh, dk, dv, p = 2, 256, 4, 256, 128, 0.2
    q = torch.randn(B, T, nh, dk, device=DEV).to(torch.bfloat16).requires_grad_()
    k = torch.randn(B, T, nh, dk, device=DEV).to(torch.bfloat16).requires_grad_()
    v = torch.randn(B, T, nh, dv, device=DEV).to(torch.bfloat16).requires_grad_()
    seed = torch.tensor([12345], dtype=torch.int64, device=DEV)
    scale = 1.0 / math.sqrt(dk)
    o = ops.causal_attention(q, k, v, scale, p, seed)
    g = torch.randn_like(o)
    o.backward(g)
    keep = dropout_keep_mask(B, nh, T, T, p, seed)
    qr, kr, vr = [t.detach().float().requires_grad_() for t in (q, k, v)]
    sc = (qr.transpose(1, 2) @ kr.transpose(1, 2).transpose(-1, -2)) * scale
    sc = sc.masked_fill(~torch.ones(T, T, dtype=torch.bool, device=DEV).tril(), float("-inf"))
    pr = torch.softmax(sc, dim=-1) * keep / (1 - p)
    orf = (pr @ vr.transpose(1, 2)).transpose(1, 2)
    orf.backward(g.float())
    assert rel_err(o, orf) < 1e-2
    assert rel_err(q.grad, qr.grad) < 2e-2 and rel_err(k.grad, kr.grad) < 2e-2 and rel_err(v.grad, vr.grad) < 2e-2


def _attn_check_chunked(q, k, v, g, o, scale, heads_per_chunk=8):
    """fp32 reference of output and gradients, computed per batch element and group of heads (the score matrices
    of the large shapes do not fit at once): the worst relative errors."""
    B, T, nh, hs = q.shape
    nkv = k.shape[2]
    G = nh // nkv
    worst = [0.0, 0.0, 0.0, 0.0]
    dk_ref = torch.zeros_like(k, dtype=torch.float32)
    dv_ref = torch.zeros_like(v, dtype=torch.float32)
    dq_ref = torch.zeros_like(q, dtype=torch.float32)
    o_ref = torch.zeros_like(o, dtype=torch.float32)
    for b in range(B):
        for h0 in range(0, nh, heads_per_chunk):
            hsl = slice(h0, h0 + heads_per_chunk)
            ksl = slice(h0 // G, (h0 + heads_per_chunk + G - 1) // G)
            qr = q[b:b + 1, :, hsl].detach().float().requires_grad_()
            kr = k[b:b + 1, :, ksl].detach().float().requires_grad_()
            vr = v[b:b + 1, :, ksl].detach().float().requires_grad_()
            orf = _attn_ref(qr, kr, vr, scale)
            orf.backward(g[b:b + 1, :, hsl].float())
            o_ref[b:b + 1, :, hsl] = orf.detach()
            dq_ref[b:b + 1, :, hsl] = qr.grad
            dk_ref[b:b + 1, :, ksl] += kr.grad
            dv_ref[b:b + 1, :, ksl] += vr.grad
    return rel_err(o, o_ref), rel_err(q.grad, dq_ref), rel_err(k.grad, dk_ref), rel_err(v.grad, dv_ref)


@pytest.mark.parametrize("B,T,nh,nkv,hs", [
    (64, 1024, 12, 12, 64),     # the headline's attention at micro-batch 64: 768 (batch, head) pairs, XCD-grouped order
    (16, 2048, 32, 8, 128),     # 1.3B-class heads with grouped keys at T = 2048
])
def test_flash_attention_at_benchmarked_shapes(B, T, nh, nkv, hs):
    from distributed_pytorch_amd import ops
    torch.manual_seed(20)
    q = torch.randn(B, T, nh, hs, device=DEV).to(torch.bfloat16).requires_grad_()
    k = torch.randn(B, T, nkv, hs, device=DEV).to(torch.bfloat16).requires_grad_()
    v = torch.randn(B, T, nkv, hs, device=DEV).to(torch.bfloat16).requires_grad_()
    scale = 1.0 / math.sqrt(hs)
    o = ops.causal_attention(q, k, v, scale)
    g = torch.randn_like(o)
    o.backward(g)
    G = nh // nkv
    eo, eq, ek, ev = _attn_check_chunked(q, k, v, g, o, scale, heads_per_chunk=max(G, 4))
    assert eo < 1e-2 and eq < 2e-2 and ek < 2e-2 and ev < 2e-2, (eo, eq, ek, ev)


@pytest.mark.parametrize("hs", [64, 128])
def test_flash_attention_underflowing_and_mixed_first_tiles(hs):
    """The forward keeps a fixed reference point until a tile's probabilities leave [1e-9, 1e9]. Rows whose every score
    in the first key tile is far below zero (k = -4 q: the first tile underflows and has to be redone with a row
    maximum), rows with a huge early score (overflow), and ordinary rows, side by side in the same 32-row blocks."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(21)
    B, T, nh = 2, 256, 4
    q = torch.randn(B, T, nh, hs, device=DEV)
    k = torch.randn(B, T, nh, hs, device=DEV)
    v = torch.randn(B, T, nh, hs, device=DEV)
    # head 0: every key opposes every query direction -> all scores strongly negative (underflow everywhere)
    base = torch.randn(hs, device=DEV)
    base = base / base.norm() * math.sqrt(hs)          # scores of head 0: about -4 sqrt(hs) (e^-32 and below)
    q[:, :, 0] = base + 0.05 * torch.randn(B, T, hs, device=DEV)
    k[:, :, 0] = -4.0 * q[:, :, 0]
    # head 1: odd rows underflow against the first 64 keys, even rows see one very large score among them
    k[:, :64, 1] = -4.0 * q[:, 65:66, 1]
    q[:, 1::2, 1] = q[:, 65:66, 1]
    q[:, 0::2, 1] = -1.5 * q[:, 65:66, 1]              # +6 sqrt(hs): e^48 and above
    q, k, v = [t.to(torch.bfloat16).requires_grad_() for t in (q, k, v)]
    scale = 1.0 / math.sqrt(hs)
    o = ops.causal_attention(q, k, v, scale)
    g = torch.randn_like(o)
    o.backward(g)
    assert torch.isfinite(o.float()).all() and torch.isfinite(q.grad.float()).all()
    qr, kr, vr = [t.detach().float().requires_grad_() for t in (q, k, v)]
    orf = _attn_ref(qr, kr, vr, scale)
    orf.backward(g.float())
    assert rel_err(o, orf) < 1e-2
    assert rel_err(q.grad, qr.grad) < 3e-2 and rel_err(k.grad, kr.grad) < 3e-2 and rel_err(v.grad, vr.grad) < 2e-2


def test_flash_attention_fused_qkv_and_spike():
    """Fused-QKV entry point + a row whose max jumps late (exercises the online-softmax rescale)."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(9)
    B, T, nh, nkv, hs = 2, 512, 6, 3, 64
    qkv = torch.randn(B, T, (nh + 2 * nkv) * hs, device=DEV)
    q, k, v = ops.split_qkv(qkv, nh, nkv)
    k[0, 400, 0] = q[0, 450, 0] * 4.0     # a late, very large score for query 450 of head 0/1
    qkv = qkv.to(torch.bfloat16).requires_grad_()
    o = ops.causal_self_attention_qkv(qkv, nh, nkv)
    g = torch.randn_like(o)
    o.backward(g)
    ref_in = qkv.detach().float().requires_grad_()
    qr, kr, vr = ops.split_qkv(ref_in, nh, nkv)
    orf = _attn_ref(qr, kr, vr, 1.0 / math.sqrt(hs))
    orf.backward(g.float())
    assert rel_err(o, orf) < 1e-2
    assert rel_err(qkv.grad, ref_in.grad) < 2e-2


@pytest.mark.parametrize("M,N,K,bias", [(4096, 768, 3072, False), (32768, 2304, 768, True), (1000, 128, 256, True),
                                        (8192, 768, 768, True), (300, 256, 128, False)])
def test_wgrad_gemm(M, N, K, bias):
    from distributed_pytorch_amd import ops
    torch.manual_seed(10)
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    dw, db = ops.weight_grad(dy, x, bias, version=1)
    want = dy.float().t() @ x.float()
    assert rel_err(dw, want) < 5e-3
    if bias:
        assert rel_err(db, dy.float().sum(0)) < 5e-3
    else:
        assert db is None


def test_linear_autograd_matches_reference():
    from distributed_pytorch_amd import ops
    torch.manual_seed(11)
    x = torch.randn(4, 256, 768, device=DEV).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(2304, 768, device=DEV) * 0.02).to(torch.bfloat16).requires_grad_()
    b = torch.randn(2304, device=DEV).to(torch.bfloat16).requires_grad_()
    y = ops.linear(x, w, b)
    g = torch.randn_like(y)
    y.backward(g)
    xr, wr, br = [t.detach().float().requires_grad_() for t in (x, w, b)]
    yr = F.linear(xr, wr, br)
    yr.backward(g.float())
    assert rel_err(y, yr) < 1e-2 and rel_err(x.grad, xr.grad) < 1e-2
    assert rel_err(w.grad, wr.grad) < 1e-2 and rel_err(b.grad, br.grad) < 1e-2


@pytest.mark.parametrize("M,N,K", [(4096, 768, 3072), (32768, 2304, 768), (1024, 200, 256), (8192, 50304, 768),
                                   (320, 128, 512), (2048, 2048, 8192), (4096, 1152, 768), (2048, 6144, 2048),
                                   (1088, 384, 256)])
def test_wgrad2_gemm(M, N, K):
    """LDS-DMA weight-gradient kernels: output heights that are not a multiple of the tile (a half-empty last row of
    256 x 256 tiles included), a wide grid of tiles (numbered in blocks), one-split and many-split products, a row
    count that is not a multiple of the ring's depth, and the accumulating form (tiles summed whole or in parts)."""
    from distributed_pytorch_amd import ops
    from distributed_pytorch_amd._ext import ops as hip
    torch.manual_seed(12)
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    assert ops.linear.__module__ and ops.wgrad_supported is not None
    dw = hip().wgrad2(dy, x)
    want = dy.float().t() @ x.float()
    assert rel_err(dw, want) < 5e-3
    ws = torch.ones(N, K, device=DEV)
    hip().wgrad2_accum(dy, x, ws)
    assert rel_err(ws, want + 1.0) < 5e-3


@pytest.mark.parametrize("M,N,dtype", [(32768, 2304, torch.bfloat16), (1000, 768, torch.bfloat16), (77, 520, torch.float32)])
def test_colsum(M, N, dtype):
    from distributed_pytorch_amd._ext import ops as hip
    torch.manual_seed(13)
    x = torch.randn(M, N, device=DEV).to(dtype)
    got = hip().colsum(x)
    assert rel_err(got, x.float().sum(0)) < 1e-4
    view = x[:, : N // 2 // 8 * 8]          # strided rows
    assert rel_err(hip().colsum(view), view.float().sum(0)) < 1e-4


@pytest.mark.parametrize("rows,N,K,kn", [(512, 256, 128, False), (1024, 3072, 768, False), (1024, 768, 3072, True),
                                         (384, 200, 192, False), (384, 264, 128, True)])
def test_grouped_gemm_forms(rows, N, K, kn):
    """Every 128-row block multiplies with its own group's matrix; out-of-range groups are skipped."""
    ops = _ops()
    torch.manual_seed(0)
    G = 3
    a = torch.randn(rows, K, device=DEV, dtype=torch.bfloat16)
    ws = [torch.randn((K, N) if kn else (N, K), device=DEV, dtype=torch.bfloat16) * 0.05 for _ in range(G)]
    nb = rows // 128
    bg = torch.tensor([(i * 7 + 1) % (G + 1) for i in range(nb)], dtype=torch.int32, device=DEV)  # value G = skip
    ptrs = torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64, device=DEV)
    c = ops.grouped_gemm(a, ptrs, ws[0].shape[0], ws[0].shape[1], kn, bg)
    for i in range(nb):
        g = int(bg[i])
        if g >= G:
            continue
        blk = a[i * 128:(i + 1) * 128].float()
        ref = blk @ (ws[g].float() if kn else ws[g].float().t())
        assert rel_err(c[i * 128:(i + 1) * 128], ref) < 6e-3, (i, g)


def test_grouped_wgrad_segments():
    ops = _ops()
    torch.manual_seed(0)
    N, K = 264, 256
    seg_rows = torch.tensor([256, 0, 128, 1024], dtype=torch.int64, device=DEV)
    seg_start = torch.cumsum(seg_rows, 0) - seg_rows
    M = int(seg_rows.sum()) + 256            # trailing rows belong to no segment
    dy = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    out = ops.grouped_wgrad(dy, x, seg_start, seg_rows)
    for g in range(4):
        s, r = int(seg_start[g]), int(seg_rows[g])
        ref = dy[s:s + r].float().t() @ x[s:s + r].float()
        if r == 0:
            assert out[g].abs().max().item() == 0
        else:
            assert rel_err(out[g], ref) < 1e-5


@pytest.mark.parametrize("nt", ["1", "0"])
@pytest.mark.parametrize("act,E,k,Ntok,C,U", [("gelu", 8, 2, 4096, 768, 3072), ("swiglu", 3, 2, 1000, 256, 384),
                                              ("gelu", 4, 1, 77, 128, 128), ("swiglu", 8, 2, 4096, 768, 2048),
                                              ("relu", 4, 2, 1000, 256, 384), ("silu", 3, 1, 300, 128, 256)])
def test_grouped_experts_fwd_bwd(act, E, k, Ntok, C, U, nt, monkeypatch):
    # nt = "1": the experts on the grouped mode of the NT GEMM (256-row blocks); "0": the 128-row grouped kernels
    monkeypatch.setenv("DPA_MOE_NT", nt)
    """The padded grouped path equals the dense-over-experts fp32 statement, gradients included
    (one expert receives no token at all in the small cases)."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(1)
    fan = 2 * U if act == "swiglu" else U
    x = torch.randn(Ntok, C, device=DEV, dtype=torch.bfloat16)
    logits = torch.randn(Ntok, E, device=DEV)
    if E <= 4:
        logits[:, E - 1] = -1e9           # nobody routes to the last expert
    gates, idx = torch.topk(logits, k, dim=1)
    gates = torch.softmax(gates, dim=1).to(torch.bfloat16)
    fcs = [(torch.randn(fan, C, device=DEV) * 0.04).to(torch.bfloat16) for _ in range(E)]
    projs = [(torch.randn(C, U, device=DEV) * 0.04).to(torch.bfloat16) for _ in range(E)]
    dout = torch.randn(Ntok, C, device=DEV, dtype=torch.bfloat16)

    xr = x.float().requires_grad_()
    gr = gates.float().requires_grad_()
    fr = [w.float().requires_grad_() for w in fcs]
    pr = [w.float().requires_grad_() for w in projs]
    ref = ops.grouped_experts_eager(xr, idx, gr, fr, pr, act)
    ref.backward(dout.float())

    xh = x.clone().requires_grad_()
    gh = gates.clone().requires_grad_()
    fh = [w.clone().requires_grad_() for w in fcs]
    ph = [w.clone().requires_grad_() for w in projs]
    assert ops.grouped_supported(xh, fh, ph, act)
    # the padded buffers are allocated uninitialised and only their pair-less rows are zeroed: hand the allocator
    # blocks full of NaN, so that a padding row nobody zeroed shows up in the weight gradients
    poison = torch.full((48 << 20,), float("nan"), device=DEV, dtype=torch.bfloat16)
    del poison
    out = ops.grouped_experts(xh, idx, gh, fh, ph, act)
    poison = torch.full((48 << 20,), float("nan"), device=DEV, dtype=torch.bfloat16)
    del poison
    out.backward(dout)
    assert torch.isfinite(out).all() and torch.isfinite(xh.grad).all()
    assert rel_err(out, ref) < 1.5e-2
    assert rel_err(xh.grad, xr.grad) < 1.5e-2
    assert rel_err(gh.grad, gr.grad) < 2e-2
    for e in range(E):
        if fr[e].grad.abs().max() == 0:
            assert fh[e].grad.abs().max() == 0 and ph[e].grad.abs().max() == 0
            continue
        assert rel_err(fh[e].grad, fr[e].grad) < 1.5e-2, e
        assert rel_err(ph[e].grad, pr[e].grad) < 1.5e-2, e


@pytest.mark.parametrize("N,E,k,with_bias", [(4096, 8, 2, False), (1000, 64, 8, True), (77, 3, 1, True), (16384, 16, 4, False)])
def test_moe_topk_gate_matches_torch(N, E, k, with_bias):
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    logits = torch.randn(N, E, device=DEV) * 2
    bias = torch.randn(E, device=DEV) if with_bias else None
    dg = torch.randn(N, k, device=DEV)
    lr = logits.clone().requires_grad_()
    biased = lr if bias is None else lr + bias
    _, idx_ref = torch.topk(biased, k, dim=1)
    g_ref = F.softmax(torch.gather(lr, 1, idx_ref), dim=1)
    g_ref.backward(dg)
    lh = logits.clone().requires_grad_()
    idx, g = ops.topk_gate(lh, bias, k)
    g.backward(dg)
    assert idx.dtype == torch.int64 and torch.equal(idx, idx_ref)     # random floats: no ties
    assert torch.allclose(g, g_ref, atol=1e-6, rtol=1e-5)
    assert torch.allclose(lh.grad, lr.grad, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("block", [128, 256])
@pytest.mark.parametrize("N,E,k", [(4096, 8, 2), (1000, 64, 8), (77, 3, 1), (65536, 8, 2), (5, 4, 2)])
def test_moe_plan_matches_sort_based_layout(N, E, k, block):
    """The histogram / scan / placement kernels produce exactly the stable-sort layout, for both row-block sizes."""
    from distributed_pytorch_amd.ops import moe
    ops = _ops()
    torch.manual_seed(0)
    idx = torch.stack([torch.randperm(E, device=DEV)[:k] for _ in range(min(N, 512))])
    idx = idx.repeat((N + idx.shape[0] - 1) // idx.shape[0], 1)[:N].contiguous()
    if E > 2:
        idx[idx == E - 1] = 0                  # leave one expert empty
    rows = moe.padded_rows(N * k, E, block)
    got = ops.moe_plan(idx.reshape(-1), E, rows, block)
    import os
    os.environ["DPA_FORCE_EAGER"] = "1"
    try:
        want = moe.plan(idx, E, block)
    finally:
        del os.environ["DPA_FORCE_EAGER"]
    assert torch.equal(got[0], want[0])                       # row_of_pair
    assert torch.equal(got[1], want[1])                       # block_group
    assert torch.equal(got[2], want[2]) and torch.equal(got[3], want[3])
    assert torch.equal(got[4], want[5])                       # counts
    assert got[0].unique().numel() == N * k


@pytest.mark.parametrize("M,N,K", [(32768, 2304, 768), (4096, 768, 768), (1024, 256, 256), (2048, 200, 256)])
def test_wgrad_with_fused_bias_sums(M, N, K):
    """Weight gradient and bias gradient from one op: column sums come out of the 256x256 kernel
    where it applies (first three shapes) and from the column-sum kernel otherwise."""
    ops = _ops()
    torch.manual_seed(13)
    dy = (torch.randn(M, N, device=DEV) + 0.1).to(torch.bfloat16)
    x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    slabs, db = ops.wgrad2_slabs_bias(dy, x)
    assert db.dtype == torch.float32 and db.shape == (N,)
    assert rel_err(slabs.sum(0), dy.float().t() @ x.float()) < 1e-5
    assert rel_err(db, dy.float().sum(0)) < 1e-5


@pytest.mark.parametrize("N,E,k,C", [(4096, 8, 2, 768), (77, 4, 1, 128), (1000, 16, 4, 256)])
def test_dense_plan_scatter_and_combine(N, E, k, C):
    """Unpadded expert-sorted order (the expert-parallel send order) and the differentiable row
    scatter / gate-weighted combine built on it, against their PyTorch statements."""
    import os
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    idx = torch.stack([torch.randperm(E, device=DEV)[:k] for _ in range(N)])
    rop, counts = ops.plan_dense(idx, E)
    os.environ["DPA_FORCE_EAGER"] = "1"
    try:
        rop_ref, counts_ref = ops.plan_dense(idx, E)
    finally:
        del os.environ["DPA_FORCE_EAGER"]
    assert torch.equal(rop, rop_ref) and torch.equal(counts, counts_ref)
    x = torch.randn(N, C, device=DEV, dtype=torch.bfloat16)
    gates = torch.softmax(torch.randn(N, k, device=DEV), 1).to(torch.bfloat16)
    dout = torch.randn(N, C, device=DEV, dtype=torch.bfloat16)

    def run(eager):
        if eager:
            os.environ["DPA_FORCE_EAGER"] = "1"
        try:
            xi = x.clone().float().requires_grad_() if eager else x.clone().requires_grad_()
            gi = gates.clone().float().requires_grad_() if eager else gates.clone().requires_grad_()
            xs = ops.scatter_rows(xi, rop, k, N * k)
            out = ops.gather_combine(xs * 2.0, rop, gi, k)
            out.backward(dout.float() if eager else dout)
            return out, xi.grad, gi.grad
        finally:
            os.environ.pop("DPA_FORCE_EAGER", None)

    got, want = run(False), run(True)
    for a, b in zip(got, want):
        assert rel_err(a, b) < 1.5e-2


@pytest.mark.parametrize("N,W,k,C,cap", [(2048, 4, 2, 768, 512), (600, 2, 2, 128, 1000), (1000, 8, 1, 256, 64)])
def test_scatter_into_capacity_layout_with_dropped_pairs(N, W, k, C, cap):
    """The static-capacity send layout of the expert-parallel exchange: ``cap`` rows per destination, pairs past a
    destination's capacity carry row -1 (not written, nothing combined, zero gate gradient), only the rows no pair
    fills are zeroed (the allocator hands out NaN-filled blocks here). Forward and backward against the PyTorch
    statement of the same functions."""
    import os
    from distributed_pytorch_amd import ops
    torch.manual_seed(1)
    dest = torch.randint(0, W, (N * k,), device=DEV)
    dest[: N * k // 3] = 0                                   # one busy destination, so that some capacities overflow
    order = torch.argsort(dest, stable=True)
    place = torch.empty_like(order)
    counts = torch.zeros(W, dtype=torch.int64, device=DEV).scatter_add_(0, dest, torch.ones_like(dest))
    start = torch.cumsum(counts, 0) - counts
    place[order] = torch.arange(N * k, device=DEV) - start[dest[order]]
    row = torch.where(place < cap, dest * cap + place, torch.full_like(place, -1))
    sent = counts.clamp(max=cap)
    lanes = torch.arange(W, device=DEV)
    tables = (lanes * cap, sent.contiguous(), torch.full_like(lanes, cap))
    x = torch.randn(N, C, device=DEV, dtype=torch.bfloat16)
    gates = torch.softmax(torch.randn(N, k, device=DEV), 1).to(torch.bfloat16)
    dout = torch.randn(N, C, device=DEV, dtype=torch.bfloat16)
    poison = [torch.full((W * cap, C), float("nan"), device=DEV, dtype=torch.bfloat16) for _ in range(3)]
    del poison                                               # the caching allocator reuses these blocks below

    def run(eager):
        if eager:
            os.environ["DPA_FORCE_EAGER"] = "1"
        try:
            xi = x.clone().float().requires_grad_() if eager else x.clone().requires_grad_()
            gi = gates.clone().float().requires_grad_() if eager else gates.clone().requires_grad_()
            xs = ops.scatter_rows(xi, row, k, W * cap, tables)
            ys = xs * 2.0
            ys.retain_grad()
            out = ops.gather_combine(ys, row, gi, k, tables)
            out.backward(dout.float() if eager else dout)
            return xs.detach(), out, xi.grad, gi.grad, ys.grad
        finally:
            os.environ.pop("DPA_FORCE_EAGER", None)

    got, want = run(False), run(True)
    assert bool((row < 0).any()) == bool((counts > cap).any())   # (the second case drops nothing)
    for a, b in zip(got, want):
        assert torch.isfinite(a.float()).all()
        assert rel_err(a, b) < 1.5e-2
    valid = torch.zeros(W * cap, dtype=torch.bool, device=DEV)
    valid[row[row >= 0]] = True
    assert (got[0][~valid] == 0).all() and (got[4][~valid] == 0).all()      # padding rows: zero data, zero gradient
    assert (got[3].reshape(-1)[row < 0] == 0).all()                          # dropped pairs: zero gate gradient


@pytest.mark.parametrize("B,Tq,Tk,nh,nkv,hs,dv", [(2, 256, 256, 8, 8, 32, 32), (2, 200, 200, 4, 1, 48, 32),
                                                  (1, 128, 128, 4, 2, 96, 96), (2, 1, 130, 4, 4, 32, 32)])
def test_flash_attention_padded_head_sizes(B, Tq, Tk, nh, nkv, hs, dv):
    """Head sizes that are not native to the kernels (and a value width below the key width) run on
    them after zero padding and match the eager statement, forward and backward."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    q = torch.randn(B, Tq, nh, hs, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, Tk, nkv, hs, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, Tk, nkv, dv, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, Tq, nh, dv, device=DEV, dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(hs)
    o = ops.causal_attention(q, k, v, scale)
    o.backward(do)
    qr, kr, vr = [t.detach().float().requires_grad_() for t in (q, k, v)]
    orf = ops.causal_attention_eager(qr, kr, vr, scale)
    orf.backward(do.float())
    assert o.shape == (B, Tq, nh, dv)
    assert rel_err(o, orf) < 1e-2
    assert rel_err(q.grad, qr.grad) < 2e-2 and rel_err(k.grad, kr.grad) < 2e-2 and rel_err(v.grad, vr.grad) < 2e-2


def test_out_of_range_ids_and_targets_are_contained():
    """Token ids / labels outside the vocabulary (a corrupt shard) read as a zero embedding row, get
    no gradient and never index out of bounds; the rest of the batch is unaffected."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    V, C, B, T = 512, 256, 4, 64
    wte = torch.randn(V, C, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    idx = torch.randint(0, V, (B, T), device=DEV)
    bad = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    bad[0, 3] = bad[1, 10] = bad[3, 63] = True
    idx_bad = idx.clone()
    idx_bad[0, 3], idx_bad[1, 10], idx_bad[3, 63] = V + 7, -5, 60000
    out = ops.embedding(idx_bad, wte)
    do = torch.randn_like(out)
    out.backward(do)
    ref_w = wte.detach().float().requires_grad_()
    ref = torch.nn.functional.embedding(idx, ref_w) * (~bad)[..., None]
    ref.backward(do.float())
    assert torch.equal(out[bad], torch.zeros_like(out[bad]))
    assert rel_err(out, ref) < 1e-3 and rel_err(wte.grad, ref_w.grad) < 1e-2
    # labels: an out-of-range target is finite and leaves the other rows' loss gradient intact
    logits = torch.randn(256, V, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    tgt = torch.randint(0, V, (256,), device=DEV)
    tgt[5] = V + 100
    loss = ops.cross_entropy(logits, tgt)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(logits.grad).all()


def _masked_attention_reference(q, k, v, scale, keep, p):
    """fp32 attention with an explicit dropout keep mask [B, nh, Tq, Tk] applied to the probabilities."""
    B, Tq, nh, hs = q.shape
    Tk, nkv = k.shape[1], k.shape[2]
    qh, kh, vh = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    kh = kh.repeat_interleave(nh // nkv, dim=1)
    vh = vh.repeat_interleave(nh // nkv, dim=1)
    s = (qh @ kh.transpose(-1, -2)) * scale
    causal = torch.ones(Tq, Tk, dtype=torch.bool, device=q.device).tril(diagonal=Tk - Tq)
    s = s.masked_fill(~causal, float("-inf"))
    pr = torch.softmax(s, dim=-1) * keep.to(s.dtype) / (1.0 - p)
    return (pr @ vh).transpose(1, 2)


@pytest.mark.parametrize("B,Tq,Tk,nh,nkv,hs,p", [(2, 256, 256, 4, 4, 64, 0.1), (1, 200, 200, 8, 2, 64, 0.25),
                                                  (2, 128, 128, 4, 2, 128, 0.1), (1, 70, 198, 4, 4, 64, 0.5)])
def test_flash_attention_dropout_matches_masked_reference(B, Tq, Tk, nh, nkv, hs, p):
    """In-kernel dropout: the forward and both backward kernels regenerate one mask from the seed; the
    mask kernel exposes it, and fp32 attention with that explicit mask is the oracle."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    q = torch.randn(B, Tq, nh, hs, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, Tk, nkv, hs, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, Tk, nkv, hs, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, Tq, nh, hs, device=DEV, dtype=torch.bfloat16)
    seed = torch.tensor([1234567890123], dtype=torch.int64, device=DEV)
    scale = 1.0 / math.sqrt(hs)
    o = ops.causal_attention(q, k, v, scale, dropout_p=p, seed=seed)
    o.backward(do)
    keep = ops.dropout_keep_mask(B, nh, Tq, Tk, p, seed)
    frac = keep.float().mean().item()
    assert abs(frac - (1.0 - p)) < 0.01, frac
    # different heads and different seeds give different masks
    assert not torch.equal(keep[0, 0], keep[0, 1])
    assert not torch.equal(keep, ops.dropout_keep_mask(B, nh, Tq, Tk, p, seed + 1))
    qr, kr, vr = [t.detach().float().requires_grad_() for t in (q, k, v)]
    ref = _masked_attention_reference(qr, kr, vr, scale, keep, p)
    ref.backward(do.float())
    assert rel_err(o, ref) < 1e-2
    assert rel_err(q.grad, qr.grad) < 2e-2 and rel_err(k.grad, kr.grad) < 2e-2 and rel_err(v.grad, vr.grad) < 2e-2


def test_attention_dropout_draws_a_new_mask_per_call_and_respects_manual_seed():
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    qkv = torch.randn(2, 128, 3 * 4 * 64, device=DEV, dtype=torch.bfloat16)
    torch.manual_seed(7)
    a = ops.causal_self_attention_qkv(qkv, 4, 4, dropout_p=0.2)
    b = ops.causal_self_attention_qkv(qkv, 4, 4, dropout_p=0.2)
    torch.manual_seed(7)
    a2 = ops.causal_self_attention_qkv(qkv, 4, 4, dropout_p=0.2)
    assert not torch.equal(a, b) and torch.equal(a, a2)
    assert torch.equal(ops.causal_self_attention_qkv(qkv, 4, 4, dropout_p=0.0),
                       ops.causal_self_attention_qkv(qkv, 4, 4))


@pytest.mark.parametrize("name", ["relu", "silu", "swish", "mish", "selu", "elu", "celu", "sigmoid", "tanh", "lrelu"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_activation_map_kernels(name, dtype):
    """Every activation name of the reference map runs on its HIP kernel and matches the stock op."""
    import torch.nn.functional as F
    from distributed_pytorch_amd.ops.activation import get_activation
    eager = {"relu": F.relu, "silu": F.silu, "swish": F.silu, "mish": F.mish, "selu": F.selu, "elu": F.elu,
             "celu": F.celu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "lrelu": lambda t: F.leaky_relu(t, 0.01)}[name]
    torch.manual_seed(0)
    x = (3 * torch.randn(64, 1024, device=DEV)).to(dtype).requires_grad_()
    dy = torch.randn(64, 1024, device=DEV).to(dtype)
    y = get_activation(name)(x)
    assert y.grad_fn is not None and "ActFn" in type(y.grad_fn).__name__
    y.backward(dy)
    xr = x.detach().float().requires_grad_()
    yr = eager(xr)
    yr.backward(dy.float())
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(y, yr) < tol and rel_err(x.grad, xr.grad) < tol


@pytest.mark.parametrize("nh,nkv,hs", [(4, 4, 64), (8, 2, 128), (12, 4, 64), (12, 1, 64), (6, 3, 128)])
def test_cached_attention_reads_the_fill_level_from_the_device(nh, nkv, hs):
    """Decode attention over a preallocated cache: the number of valid keys is a device scalar, stale
    rows beyond it are never read, and changing the scalar changes the result without a new launch shape."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    B, S = 3, 512
    kc = torch.randn(B, S, nkv, hs, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(B, S, nkv, hs, device=DEV, dtype=torch.bfloat16)
    q = torch.randn(B, 1, nh, hs, device=DEV, dtype=torch.bfloat16)
    scale = 1.0 / math.sqrt(hs)
    for n in (1, 63, 64, 65, 200, 512):
        kv_len = torch.tensor([n], dtype=torch.int32, device=DEV)
        o = ops.cached_attention(q, kc, vc, kv_len, scale)
        ref = ops.causal_attention_eager(q.float(), kc[:, :n].float(), vc[:, :n].float(), scale)
        assert rel_err(o, ref) < 1e-2, n


def test_fused_sampler_top1_is_argmax_and_samples_stay_in_the_top_k():
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float32):
        logits = (3 * torch.randn(33, 50304, device=DEV)).to(dtype)
        assert torch.equal(ops.sample_fused(logits, 1.0, 1)[:, 0], logits.float().argmax(-1))
        k = 50
        kth = torch.topk(logits.float(), k).values[:, -1:]
        for _ in range(4):
            tok = ops.sample_fused(logits, 0.8, k)
            assert tok.shape == (33, 1) and tok.dtype == torch.int64
            assert (logits.float().gather(1, tok) >= kth).all()
    # a short row with k larger than the row (kernel), and an odd-sized row (stock-op fallback)
    for width in (16, 13):
        small = torch.randn(5, width, device=DEV)
        tok = ops.sample_fused(small, 1.0, 100)
        assert int(tok.min()) >= 0 and int(tok.max()) < width


@pytest.mark.parametrize("topk,temperature", [(None, 1.0), (8, 0.7)])
def test_fused_sampler_follows_the_softmax_distribution(topk, temperature):
    """Many rows with the same logits: the empirical token frequencies match softmax(top-k(logits) / T)."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(1)
    V, rows = 40, 32768
    base = 2 * torch.randn(V, device=DEV)
    logits = base.expand(rows, V).contiguous()
    x = base / temperature
    if topk is not None:
        x = x.masked_fill(x < torch.topk(x, topk).values[-1], -float("inf"))
    want = torch.softmax(x, -1)
    counts = torch.zeros(V, device=DEV)
    n_draws = 4
    for _ in range(n_draws):
        tok = ops.sample_fused(logits, temperature, topk)
        counts += torch.bincount(tok[:, 0], minlength=V).float()
    got = counts / (rows * n_draws)
    assert (got[want == 0] == 0).all()
    assert (got - want).abs().max().item() < 6e-3, (got - want).abs().max().item()
    # a fixed seed reproduces the draw, another seed changes it
    seed = torch.tensor([99], dtype=torch.int64, device=DEV)
    a = ops.sample_fused(logits, temperature, topk, seed)
    assert torch.equal(a, ops.sample_fused(logits, temperature, topk, seed))
    assert not torch.equal(a, ops.sample_fused(logits, temperature, topk, seed + 1))


def test_kv_append_writes_one_row_at_the_device_position():
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    B, S, H, hs = 3, 32, 2, 64
    kc = torch.zeros(B, S, H, hs, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    qkv = torch.randn(B, 1, (4 + 2 * H) * hs, device=DEV, dtype=torch.bfloat16)     # rows are slices of a fused projection
    _, k, v = ops.split_qkv(qkv, 4, H)
    pos = torch.tensor([5], dtype=torch.int64, device=DEV)
    ops.kv_append(kc, vc, k, v, pos)
    assert torch.equal(kc[:, 5], k[:, 0]) and torch.equal(vc[:, 5], v[:, 0])
    assert kc[:, :5].abs().sum() == 0 and kc[:, 6:].abs().sum() == 0
    before = kc.clone()
    ops.kv_append(kc, vc, k, v, torch.tensor([S], dtype=torch.int64, device=DEV))   # past the end: nothing is written
    assert torch.equal(kc, before)


# ---- NT GEMM with fused epilogues (csrc/gemm_nt.hip) ----------------------------------------------
def _gemm_ref(a, b, bias):
    v = a.float() @ b.float().t()
    return v if bias is None else v + bias.float()


@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1000, 752, 192), (4096, 2304, 768), (2600, 768, 3072),
                                   (33000, 768, 768)])
@pytest.mark.parametrize("bias", [False, True])
def test_gemm_nt_plain_and_bias(M, N, K, bias):
    torch.manual_seed(0)
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    bv = (torch.randn(N, device=DEV) * 0.5).bfloat16() if bias else None
    assert _ops().gemm_nt_supported(a, b, bv, None)
    c, aux = _ops().gemm_nt(a, b, bv, 0, None)
    assert aux is None or aux.numel() == 0
    assert rel_err(c, _gemm_ref(a, b, bv)) < 4e-3


def test_gemm_nt_strided_rows():
    """A read through a row-strided view (a column slice of a wider activation)."""
    torch.manual_seed(1)
    wide = torch.randn(2048, 1024, device=DEV).bfloat16()
    a = wide[:, 256:768]
    b = (torch.randn(384, 512, device=DEV) * 0.05).bfloat16()
    c, _ = _ops().gemm_nt(a, b, None, 0, None)
    assert rel_err(c, _gemm_ref(a, b, None)) < 4e-3


@pytest.mark.parametrize("M,N,K,wscale", [(2048, 768, 256, 0.1), (5000, 3072, 768, 0.1), (2048, 768, 256, 0.6)])
def test_gemm_nt_gelu_epilogue(M, N, K, wscale):
    """c = GELU_erf(a b^T + bias), aux = GELU_erf'(a b^T + bias), both against fp32 torch. The last case has
    pre-activations of standard deviation ~10: most of them beyond the +-4 range of the kernel's polynomial for Phi."""
    torch.manual_seed(2)
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = (torch.randn(N, K, device=DEV) * wscale).bfloat16()
    bv = (torch.randn(N, device=DEV) * 0.5).bfloat16()
    c, aux = _ops().gemm_nt(a, b, bv, 1, None)
    h = _gemm_ref(a, b, bv).requires_grad_()
    act = F.gelu(h)
    act.sum().backward()
    assert rel_err(c, act) < 4e-3
    assert rel_err(aux, h.grad) < 4e-3
    # the erf approximation itself (no bf16 rounding in the way): output error far below bf16 resolution
    assert (c.float() - act).abs().max().item() < 0.02 * act.abs().max().item()
    # element-wise: within bf16 rounding of the exact value plus the 1.3e-4 the clamped tail may be off by
    tol = act.abs() * 2.0 ** -7 + 3e-4
    assert bool(((c.float() - act.detach()).abs() <= tol).all())
    assert bool(((aux.float() - h.grad).abs() <= h.grad.abs() * 2.0 ** -7 + 3e-4).all())


@pytest.mark.parametrize("epi", [2, 3])
def test_gemm_nt_mul_add_epilogues(epi):
    torch.manual_seed(3)
    M, N, K = 3100, 1024, 320
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    x = torch.randn(M, N, device=DEV).bfloat16()
    c, _ = _ops().gemm_nt(a, b, None, epi, x)
    v = _gemm_ref(a, b, None)
    ref = v * x.float() if epi == 2 else v + x.float()
    assert rel_err(c, ref) < 4e-3


def test_gemm_nt_many_tiles_per_workgroup_is_race_free():
    """The persistent path (several output tiles per workgroup, operand stream running across tile
    boundaries behind counted waits): bitwise identical results over repeated launches and equal to the
    reference on every element."""
    torch.manual_seed(4)
    M, N, K = 65536, 768, 768
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    first, _ = _ops().gemm_nt(a, b, None, 0, None)
    ref = _gemm_ref(a, b, None)
    assert ((first.float() - ref).abs() <= 2e-2 * ref.abs() + 2e-2).all()
    for _ in range(5):
        again, _ = _ops().gemm_nt(a, b, None, 0, None)
        assert torch.equal(first, again)


def test_gemm_nt_rows_past_two_gib_run_as_row_ranges():
    """Operands whose rows span 2 GiB and more (the kernel's own byte offsets stop there) are walked in row ranges:
    add epilogue (output and auxiliary operand both re-based) and the two SwiGLU forms, checked on the rows around
    the range boundaries and at both ends."""
    o = _ops()
    torch.manual_seed(11)
    K, N = 128, 1024
    M = (1 << 20) + 512                              # M * N * 2 bytes = 2 GiB + 1 MiB -> two ranges, the second 768 rows
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = (torch.randn(N, K, device=DEV) * 0.1).bfloat16()
    x = torch.randn(M, N, device=DEV).bfloat16()
    assert o.gemm_nt_supported(a, b, None, x)
    c, _ = o.gemm_nt(a, b, None, 3, x)
    cut = ((1 << 31) - 1) // (N * 2) // 256 * 256
    rows = torch.tensor([0, 1, 255, 256, cut - 1, cut, cut + 1, cut + 255, cut + 256, M - 2, M - 1], device=DEV)
    ref = a[rows].float() @ b.float().t() + x[rows].float()
    assert rel_err(c[rows], ref) < 4e-3
    del c, x
    U, Ms = 2048, (1 << 18) + 256                    # [gate | up] rows of 8 KiB: 2 GiB at 262144 rows
    a2 = a[:Ms]
    w = (torch.randn(2 * U, K, device=DEV) / K ** 0.5).bfloat16()
    assert o.gemm_nt_pair_supported(a2, w, 4, None)
    act, h = o.gemm_nt(a2, w, None, 4, None)
    cut = ((1 << 31) - 1) // (2 * U * 2) // 256 * 256
    rows = torch.tensor([0, 255, cut - 1, cut, cut + 1, Ms - 1], device=DEV)
    hf = a2[rows].float() @ w.float().t()
    want = torch.nn.functional.silu(hf[:, :U]) * hf[:, U:]
    assert (h[rows].float() - hf).abs().max() <= 2e-2 * hf.abs().max()
    assert (act[rows].float() - want).abs().max() <= 2e-2 * want.abs().max() + 1e-3
    del act
    C = 256
    dy = torch.randn(Ms, C, device=DEV).bfloat16()
    wpt = (torch.randn(U, C, device=DEV) / C ** 0.5).bfloat16()
    assert o.gemm_nt_pair_supported(dy, wpt, 5, h)
    dh = o.gemm_nt(dy, wpt, None, 5, h)[0]
    v = dy[rows].float() @ wpt.float().t()
    g, u = h[rows].float()[:, :U], h[rows].float()[:, U:]
    sg = torch.sigmoid(g)
    assert (dh[rows][:, :U].float() - v * u * sg * (1 + g * (1 - sg))).abs().max() <= 2e-2 * (v * u).abs().max() + 1e-3
    assert (dh[rows][:, U:].float() - v * g * sg).abs().max() <= 2e-2 * (v * g).abs().max() + 1e-3


def test_small_products_go_to_the_library(monkeypatch):
    """Dispatch rule: below half a round of 256 x 256 output tiles the library serves the product (same result)."""
    import importlib
    L = importlib.import_module("distributed_pytorch_amd.ops.linear")
    a = torch.randn(4096, 768, device=DEV).bfloat16()
    w_small = (torch.randn(768, 768, device=DEV) * 0.05).bfloat16()      # 16 x 3 tiles
    w_large = (torch.randn(3072, 768, device=DEV) * 0.05).bfloat16()     # 16 x 12 tiles
    monkeypatch.setattr(L, "GEMM_MIN_TILES", 128)
    assert not L.gemm_supported(a, w_small) and L.gemm_supported(a, w_large)
    y_lib = L.linear(a, w_small)
    monkeypatch.setattr(L, "GEMM_MIN_TILES", 0)
    assert L.gemm_supported(a, w_small)
    assert rel_err(L.linear(a, w_small), y_lib.float()) < 4e-3


def test_transpose_bf16():
    x = torch.randn(3072, 768, device=DEV).bfloat16()
    assert torch.equal(_ops().transpose_bf16(x), x.t().contiguous())
    y = torch.randn(200, 72, device=DEV).bfloat16()
    assert torch.equal(_ops().transpose_bf16(y), y.t().contiguous())


def test_linear_fwd_bwd_on_own_gemm():
    """ops.linear forward / input gradient on the NT kernel (bias epilogue, transposed-weight cache)
    and the weight / bias gradients, against fp32 autograd."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(5)
    M, K, N = 4096, 768, 2304
    x = torch.randn(M, K, device=DEV).bfloat16().requires_grad_()
    w = (torch.randn(N, K, device=DEV) * 0.05).bfloat16().requires_grad_()
    bv = (torch.randn(N, device=DEV) * 0.5).bfloat16().requires_grad_()
    dy = torch.randn(M, N, device=DEV).bfloat16()
    assert ops.linear.__module__.endswith("linear")
    y = ops.linear(x, w, bv)
    y.backward(dy)
    xr, wr, br = [t.detach().float().requires_grad_() for t in (x, w, bv)]
    yr = F.linear(xr, wr, br)
    yr.backward(dy.float())
    assert rel_err(y, yr) < 4e-3
    assert rel_err(x.grad, xr.grad) < 4e-3
    assert rel_err(w.grad, wr.grad) < 6e-3
    assert rel_err(bv.grad, br.grad) < 6e-3
    # a changed weight (version counter) must not be served from the transposed-weight cache
    with torch.no_grad():
        w.mul_(2.0)
    x.grad = None
    ops.linear(x, w, bv).backward(dy)
    assert rel_err(x.grad, 2 * xr.grad) < 4e-3


def test_mlp_gelu_fused_fwd_bwd():
    """Fused feed-forward block (GELU / GELU' in the GEMM epilogues) against fp32 autograd of
    c_proj(gelu(c_fc(x)))."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(6)
    M, C, U = 4096, 768, 3072
    x = torch.randn(M, C, device=DEV).bfloat16().requires_grad_()
    w1 = (torch.randn(U, C, device=DEV) * 0.05).bfloat16().requires_grad_()
    w2 = (torch.randn(C, U, device=DEV) * 0.05).bfloat16().requires_grad_()
    dy = torch.randn(M, C, device=DEV).bfloat16()
    assert ops.mlp_gelu_supported(x, w1, w2)
    y = ops.mlp_gelu(x, w1, w2)
    y.backward(dy)
    xr, w1r, w2r = [t.detach().float().requires_grad_() for t in (x, w1, w2)]
    yr = F.linear(F.gelu(F.linear(xr, w1r)), w2r)
    yr.backward(dy.float())
    assert rel_err(y, yr) < 6e-3
    assert rel_err(x.grad, xr.grad) < 8e-3
    assert rel_err(w1.grad, w1r.grad) < 8e-3
    assert rel_err(w2.grad, w2r.grad) < 8e-3


@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("acc_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (2048, 4096, 4096)])
def test_wgrad2_direct_accumulates_into_nonzero_flat_views(with_bias, acc_dtype, M, N, K):
    """The hot-path weight-gradient op: dW (and the bias gradient) are ADDED into views of a flat
    gradient buffer that already holds a previous micro-step's gradient. The second shape is a one-split
    product (256 output tiles), which the kernel adds into the gradient itself without slabs."""
    torch.manual_seed(7)
    dy = torch.randn(M, N, device=DEV).bfloat16()
    x = torch.randn(M, K, device=DEV).bfloat16()
    flat = (torch.randn(N * K + N + 64, device=DEV) * 5).to(acc_dtype)
    before = flat.clone()
    dw_view = flat[: N * K].view(N, K)
    db_view = flat[N * K: N * K + N] if with_bias else None
    _ops().wgrad2_direct(dy, x, dw_view, db_view)
    want_w = before[: N * K].float().view(N, K) + dy.float().t() @ x.float()
    tol = 1e-2 if acc_dtype == torch.bfloat16 else 4e-3
    assert rel_err(dw_view, want_w) < tol
    if with_bias:
        want_b = before[N * K: N * K + N].float() + dy.float().sum(0)
        assert rel_err(db_view, want_b) < tol
    else:
        assert torch.equal(flat[N * K:], before[N * K:])   # nothing beyond the weight view is touched
    assert torch.equal(flat[N * K + N:], before[N * K + N:])


def test_rope_qkv_inplace_matches_eager_and_rotates_the_gradient_back():
    """In-place rotary embedding of the q / k heads of a fused projection: same values as the fp32 eager
    rotation of the slice, v untouched, and the gradient of a loss on the output is the inverse rotation."""
    from distributed_pytorch_amd.ops.rope import _rope_eager, rope_qkv_, rope_tables
    torch.manual_seed(0)
    B, T, nh, nkv, hs = 2, 96, 4, 2, 64
    W = (nh + 2 * nkv) * hs
    cos_t, sin_t = rope_tables(128, hs, device="cuda")
    base = torch.randn(B, T, W, device="cuda")
    x = base.bfloat16().requires_grad_()
    qkv = x * 1.0                                    # a tensor the op may own (not a leaf)
    out = rope_qkv_(qkv, cos_t, sin_t, 5, nh + nkv, hs)
    assert out.data_ptr() == qkv.data_ptr()          # really in place
    ref_in = base.bfloat16().float()
    want_qk = _rope_eager(ref_in[..., : (nh + nkv) * hs].reshape(B, T, nh + nkv, hs), cos_t, sin_t, 5, False)
    assert torch.allclose(out[..., : (nh + nkv) * hs].float().view(B, T, nh + nkv, hs), want_qk, atol=2e-2, rtol=2e-2)
    assert torch.equal(out[..., (nh + nkv) * hs:], base.bfloat16()[..., (nh + nkv) * hs:])
    g = torch.randn(B, T, W, device="cuda").bfloat16()
    out.backward(g.clone())                          # the op rotates the gradient it is handed in place
    want_g = _rope_eager(g.float()[..., : (nh + nkv) * hs].reshape(B, T, nh + nkv, hs), cos_t, sin_t, 5, True)
    assert torch.allclose(x.grad[..., : (nh + nkv) * hs].float().view(B, T, nh + nkv, hs), want_g, atol=2e-2, rtol=2e-2)
    assert torch.equal(x.grad[..., (nh + nkv) * hs:], g[..., (nh + nkv) * hs:])


@pytest.mark.parametrize("M,U,K", [(1024, 256, 256), (2100, 1088, 128), (4096, 3072, 768)])
def test_gemm_nt_swiglu_epilogues_match_fp32(M, U, K):
    """Forward: act = silu(x Wg^T) * (x Wu^T) with [gate | up] also returned; backward: [dgate | dup] from
    d(act) = dy Wp. Both against fp32 PyTorch on the bf16 operands (ragged rows and a ragged last tile)."""
    o = torch.ops.dpa
    torch.manual_seed(M + U)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(2 * U, K, device="cuda") / K ** 0.5).bfloat16()
    assert o.gemm_nt_pair_supported(x, w, 4, None)
    act, h = o.gemm_nt(x, w, None, 4, None)
    hf = x.float() @ w.float().t()
    want = torch.nn.functional.silu(hf[:, :U]) * hf[:, U:]
    assert act.shape == (M, U) and h.shape == (M, 2 * U)
    assert (h.float() - hf).abs().max() <= 2e-2 * hf.abs().max()
    assert (act.float() - want).abs().max() <= 2e-2 * want.abs().max() + 1e-3
    # backward form: v = dy @ wp^T (NT on the transposed projection weight), h from the forward
    C = 256
    dy = torch.randn(M, C, device="cuda").bfloat16()
    wpt = (torch.randn(U, C, device="cuda") / C ** 0.5).bfloat16()      # W_proj^T: [U, C]
    assert o.gemm_nt_pair_supported(dy, wpt, 5, h)
    dh = o.gemm_nt(dy, wpt, None, 5, h)[0]
    assert dh.shape == (M, 2 * U)
    v = dy.float() @ wpt.float().t()
    g, u = h.float()[:, :U], h.float()[:, U:]
    s = torch.sigmoid(g)
    want_dg = v * u * s * (1 + g * (1 - s))
    want_du = v * g * s
    assert (dh[:, :U].float() - want_dg).abs().max() <= 2e-2 * want_dg.abs().max() + 1e-3
    assert (dh[:, U:].float() - want_du).abs().max() <= 2e-2 * want_du.abs().max() + 1e-3


def test_mlp_swiglu_fused_matches_unfused():
    from distributed_pytorch_amd import ops
    torch.manual_seed(3)
    M, C, U = 2048, 256, 512
    x = torch.randn(4, M // 4, C, device="cuda").bfloat16().requires_grad_()
    w_fc = (torch.randn(2 * U, C, device="cuda") / C ** 0.5).bfloat16().requires_grad_()
    w_pr = (torch.randn(C, U, device="cuda") / U ** 0.5).bfloat16().requires_grad_()
    assert ops.mlp_swiglu_supported(x, w_fc, w_pr)
    y = ops.mlp_swiglu(x, w_fc, w_pr)
    gy = torch.randn_like(y)
    y.backward(gy)
    got = [y.detach().float(), x.grad.float(), w_fc.grad.float(), w_pr.grad.float()]
    xr, fr, pr = (t.detach().float().requires_grad_() for t in (x, w_fc, w_pr))
    hr = xr @ fr.t()
    yr = (torch.nn.functional.silu(hr[..., :U]) * hr[..., U:]) @ pr.t()
    yr.backward(gy.float())
    for a, b, name in zip(got, [yr.detach(), xr.grad, fr.grad, pr.grad], ["y", "dx", "dw_fc", "dw_proj"]):
        err = ((a - b).norm() / b.norm()).item()
        assert err < 2e-2, f"{name}: relative error {err:.4f}"


@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_nt_grouped_matches_per_block_reference(epi):
    """Grouped mode of the NT GEMM: every 256-row block against its own weight matrix; a block without a
    group gives zeros; GELU (+ derivative) and multiply epilogues."""
    o = torch.ops.dpa
    torch.manual_seed(epi)
    G, N, K, nblk = 3, 384, 256, 7
    rows = nblk * 256
    a = torch.randn(rows, K, device="cuda").bfloat16()
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(G)]
    ptrs = torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64, device="cuda")
    groups = torch.tensor([0, 0, 2, 1, 3, 2, -1], dtype=torch.int32, device="cuda")     # 3 and -1: no group
    aux = torch.randn(rows, N, device="cuda").bfloat16() if epi == 2 else None
    c, extra = o.gemm_nt_grouped(a, ptrs, N, K, epi, aux, groups)
    for b, g in enumerate(groups.tolist()):
        blk = slice(256 * b, 256 * (b + 1))
        v = a[blk].float() @ ws[g].float().t() if 0 <= g < G else torch.zeros(256, N, device="cuda")
        if epi == 1:
            want = torch.nn.functional.gelu(v)
            vv = v.clone().requires_grad_()
            torch.nn.functional.gelu(vv).sum().backward()
            assert (extra[blk].float() - vv.grad).abs().max() <= 2e-2
        elif epi == 2:
            want = v * aux[blk].float()
        else:
            want = v
        assert (c[blk].float() - want).abs().max() <= 2e-2 * want.abs().max() + 1e-3, (b, g)


def test_gemm_nt_grouped_rows_past_two_gib():
    """Grouped mode over more than 2 GiB of output rows: the row ranges carry their slice of the block table."""
    o = torch.ops.dpa
    torch.manual_seed(9)
    G, N, K = 3, 1024, 128
    nblk = (1 << 12) + 3                                   # 4099 blocks of 256 rows: 2 GiB + 1.5 MiB of output
    rows = nblk * 256
    a = torch.randn(rows, K, device="cuda").bfloat16()
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16() for _ in range(G)]
    ptrs = torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64, device="cuda")
    groups = (torch.arange(nblk, device="cuda", dtype=torch.int32) * 7 + 1) % (G + 1)    # value G: no group
    c, _ = o.gemm_nt_grouped(a, ptrs, N, K, 0, None, groups)
    cut = ((1 << 31) - 1) // (N * 2) // 256                # first block of the second range
    for b in (0, 1, cut - 1, cut, cut + 1, nblk - 1):
        g = int(groups[b])
        blk = slice(256 * b, 256 * (b + 1))
        want = a[blk].float() @ ws[g].float().t() if g < G else torch.zeros(256, N, device="cuda")
        assert (c[blk].float() - want).abs().max() <= 2e-2 * want.abs().max() + 1e-3, (b, g)


def test_grouped_wgrad_on_the_256_kernel_matches():
    """Grouped mode of the dense 256 x 256 weight-gradient kernel (equal row chunks over the segments, combined
    atomically): per-segment dy^T x with an empty, a short and a multi-chunk segment, against fp32."""
    o = torch.ops.dpa
    torch.manual_seed(5)
    seg = torch.tensor([1024, 0, 256, 9216], dtype=torch.int64)
    start = torch.cumsum(seg, 0) - seg
    rows, N, K = int(seg.sum()) + 256, 512, 256          # 256 trailing padding rows belong to no segment
    dy = torch.randn(rows, N, device="cuda").bfloat16()
    x = torch.randn(rows, K, device="cuda").bfloat16()
    o.grouped_wgrad_use_wgrad3(True, True)      # second flag: whatever the row count
    try:
        got = o.grouped_wgrad(dy, x, start.cuda(), seg.cuda())
    finally:
        o.grouped_wgrad_use_wgrad3(True, False)
    for g in range(4):
        s, n = int(start[g]), int(seg[g])
        want = dy[s:s + n].float().t() @ x[s:s + n].float()
        assert (got[g] - want).abs().max() <= 1e-3 * max(want.abs().max().item(), 1.0), g


@pytest.mark.parametrize("B,T,nh,nkv,hs,C,pos0,bias", [(4, 128, 4, 2, 64, 256, 0, True), (3, 256, 6, 6, 128, 768, 0, False),
                                                       (5, 128, 12, 4, 64, 768, 100, True), (2, 1024, 12, 12, 64, 768, 0, True)])
def test_gemm_nt_rope_epilogue_matches_fp32_rotation(B, T, nh, nkv, hs, C, pos0, bias):
    """The qkv projection with the rotary embedding in its epilogue against an fp32 product followed by the
    eager fp32 rotation of the q and k heads; the v columns are the plain product."""
    from distributed_pytorch_amd.ops.rope import _rope_eager, rope_tables
    torch.manual_seed(0)
    W = (nh + 2 * nkv) * hs
    cos_t, sin_t = rope_tables(pos0 + T + 3, hs, device="cuda")
    x = torch.randn(B * T, C, device="cuda").bfloat16()
    w = (torch.randn(W, C, device="cuda") / math.sqrt(C)).bfloat16()
    b = torch.randn(W, device="cuda").bfloat16() if bias else None
    got = _ops().gemm_nt_rope(x, w, b, cos_t, sin_t, pos0, T, (nh + nkv) * hs, hs)
    ref = x.float() @ w.float().t() + (b.float() if bias else 0.0)
    rq = _rope_eager(ref[:, : (nh + nkv) * hs].reshape(B, T, nh + nkv, hs), cos_t, sin_t, pos0, False)
    ref = torch.cat([rq.reshape(B * T, -1), ref[:, (nh + nkv) * hs:]], dim=1)
    assert rel_err(got, ref) < 6e-3
    assert torch.allclose(got.float(), ref, atol=3e-2, rtol=2e-2)
    plain = _ops().gemm_nt(x, w, b, 0, None)[0]
    assert torch.equal(got[:, (nh + nkv) * hs:], plain[:, (nh + nkv) * hs:])


@pytest.mark.parametrize("nh,nkv,hs", [(4, 2, 64), (6, 6, 128)])
def test_qkv_rope_attention_fused_matches_separate_passes(nh, nkv, hs):
    """Projection + rotary + attention with the rotation inside the GEMM epilogue and its inverse inside
    the attention backward stores, against the same block with separate rotary passes, and against fp32."""
    from distributed_pytorch_amd import ops
    from distributed_pytorch_amd.ops.rope import _rope_eager, rope_tables
    torch.manual_seed(1)
    B, T, C = 4, 256, nh * hs
    W = (nh + 2 * nkv) * hs
    cos_t, sin_t = rope_tables(T, hs, device="cuda")
    x0 = torch.randn(B, T, C, device="cuda").bfloat16()
    w0 = (torch.randn(W, C, device="cuda") / math.sqrt(C)).bfloat16()
    b0 = (torch.randn(W, device="cuda") * 0.1).bfloat16()
    do = torch.randn(B, T, nh, hs, device="cuda").bfloat16()
    assert ops.qkv_rope_attention_supported(x0, w0, b0, cos_t, sin_t, 0, hs)

    def run(kind):
        dt = torch.float32 if kind == "fp32" else torch.bfloat16
        x, w, b = [t.detach().to(dt).requires_grad_() for t in (x0, w0, b0)]
        if kind == "fused":
            o = ops.qkv_rope_attention(x, w, b, cos_t, sin_t, 0, nh, nkv)
        elif kind == "separate":
            qkv = ops.rope_qkv_(ops.linear(x, w, b), cos_t, sin_t, 0, nh + nkv, hs)
            o = ops.causal_self_attention_qkv(qkv, nh, nkv)
        else:
            qkv = F.linear(x, w, b)
            qk = _rope_eager(qkv[..., : (nh + nkv) * hs].reshape(B, T, nh + nkv, hs), cos_t, sin_t, 0, False)
            q, k, v = qk[:, :, :nh], qk[:, :, nh:], qkv[..., (nh + nkv) * hs:].reshape(B, T, nkv, hs)
            o = ops.causal_attention_eager(q, k, v, 1.0 / math.sqrt(hs))
        o.backward(do.to(dt))
        return o, x.grad, w.grad, b.grad

    fused, sep, ref = run("fused"), run("separate"), run("fp32")
    for name, f, s, r in zip(("o", "dx", "dw", "db"), fused, sep, ref):
        assert rel_err(f, r) < 2e-2, name
        assert rel_err(f, s) < 2e-2, name
        assert rel_err(f, r) < 1.5 * rel_err(s, r) + 2e-3, name   # one rounding fewer, never less accurate


@pytest.mark.parametrize("B,N,K", [(1, 768, 768), (8, 2304, 768), (16, 768, 3072), (3, 1024, 256), (5, 50304, 768),
                                   (8, 2048, 8192), (7, 8192, 2048), (17, 768, 768), (37, 2304, 768), (64, 768, 3072)])
@pytest.mark.parametrize("form", ["plain", "ln_bias", "ln_gelu", "bias_residual", "ln_swiglu"])
def test_decode_linear_forms_match_fp32(B, N, K, form):
    """The one-launch decode-step linear layer (LayerNorm prologue, bias / GELU / SwiGLU / residual
    epilogues) against the same chain of stock fp32 ops."""
    from distributed_pytorch_amd import ops
    if form == "ln_swiglu" and N > 4096:
        pytest.skip("no gated head")
    torch.manual_seed(0)
    pair = form == "ln_swiglu"
    x = (torch.randn(B, K, device="cuda") * 1.5 + 0.3).bfloat16()
    w = (torch.randn(N * (2 if pair else 1), K, device="cuda") / math.sqrt(K)).bfloat16()
    lw = (torch.randn(K, device="cuda") * 0.2 + 1).bfloat16()
    lb = (torch.randn(K, device="cuda") * 0.1).bfloat16()
    bias = (torch.randn(w.shape[0], device="cuda") * 0.2).bfloat16()
    res = torch.randn(B, N, device="cuda").bfloat16()
    ln = (lw, lb, 1e-5) if form.startswith("ln") else None
    b = bias if "bias" in form else None
    r = res if "residual" in form else None
    act = {"ln_gelu": ops.ACT_GELU, "ln_swiglu": ops.ACT_SWIGLU}.get(form, ops.ACT_NONE)
    if ln is not None and K > 2048:
        assert not ops.decode_linear_supported(x, w, act, True)   # the prologue keeps a row in registers
        return
    assert ops.decode_linear_supported(x, w, act, ln is not None)
    got = ops.decode_linear(x, ln, w, b, r, act)
    lnf = None if ln is None else (lw.float(), lb.float(), 1e-5)
    want = ops.decode_linear_eager(x.float(), lnf, w.float(), None if b is None else b.float(),
                                   None if r is None else r.float(), act)
    assert got.shape == (B, N) and got.dtype == torch.bfloat16
    assert rel_err(got, want) < 1e-2
    assert torch.allclose(got.float(), want, atol=4e-2, rtol=3e-2)


@pytest.mark.parametrize("batch", [4, 35])
def test_decode_linear_writes_new_key_value_rows_into_the_cache(batch):
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    B, S, nh, nkv, hs, K = batch, 32, 8, 2, 64, 512
    W = (nh + 2 * nkv) * hs
    x = torch.randn(B, K, device="cuda").bfloat16()
    w = (torch.randn(W, K, device="cuda") / math.sqrt(K)).bfloat16()
    bias = (torch.randn(W, device="cuda") * 0.1).bfloat16()
    kc = torch.randn(B, S, nkv, hs, device="cuda").bfloat16()
    vc = torch.randn(B, S, nkv, hs, device="cuda").bfloat16()
    k0, v0 = kc.clone(), vc.clone()
    pos = torch.tensor([7], dtype=torch.int64, device="cuda")
    full = ops.decode_linear(x, None, w, bias)
    q = ops.decode_linear(x, None, w, bias, kv=(kc, vc, pos, nh * hs))
    assert q.shape == (B, nh * hs) and torch.equal(q, full[:, : nh * hs])
    assert torch.equal(kc[:, 7].reshape(B, -1), full[:, nh * hs: (nh + nkv) * hs])
    assert torch.equal(vc[:, 7].reshape(B, -1), full[:, (nh + nkv) * hs:])
    keep = torch.ones(S, dtype=torch.bool, device="cuda")
    keep[7] = False
    assert torch.equal(kc[:, keep], k0[:, keep]) and torch.equal(vc[:, keep], v0[:, keep])   # nothing else is touched


@pytest.mark.parametrize("batch,hs", [(3, 64), (20, 128)])
def test_decode_linear_rotates_query_and_key_heads(batch, hs):
    """The rotary epilogue of the decode-step qkv projection: query columns returned rotated, key rows cached rotated,
    value rows untouched, against the plain projection followed by the fp32 rotation."""
    from distributed_pytorch_amd import ops
    from distributed_pytorch_amd.ops.rope import rope_tables
    torch.manual_seed(2)
    B, S, nh, nkv, K = batch, 16, 4, 2, 512
    W = (nh + 2 * nkv) * hs
    x = torch.randn(B, K, device="cuda").bfloat16()
    w = (torch.randn(W, K, device="cuda") / math.sqrt(K)).bfloat16()
    bias = (torch.randn(W, device="cuda") * 0.1).bfloat16()
    lnp = ((torch.randn(K, device="cuda") * 0.2 + 1).bfloat16(), (torch.randn(K, device="cuda") * 0.1).bfloat16(), 1e-5)
    cos_t, sin_t = rope_tables(64, hs, device="cuda")
    p = 11
    kc = torch.zeros(B, S, nkv, hs, device="cuda").bfloat16()
    vc = torch.zeros_like(kc)
    pos = torch.tensor([5], dtype=torch.int64, device="cuda")
    q = ops.decode_linear(x, lnp, w, bias, kv=(kc, vc, pos, nh * hs), rope=(cos_t[p].contiguous(), sin_t[p].contiguous(), (nh + nkv) * hs, hs))
    full = ops.decode_linear_eager(x.float(), (lnp[0].float(), lnp[1].float(), 1e-5), w.float(), bias.float(), None, ops.ACT_NONE)
    qk = full[:, : (nh + nkv) * hs].view(B, nh + nkv, hs // 2, 2)
    c, s_ = cos_t[p].view(1, 1, -1), sin_t[p].view(1, 1, -1)
    rot = torch.stack((qk[..., 0] * c - qk[..., 1] * s_, qk[..., 0] * s_ + qk[..., 1] * c), dim=-1).view(B, -1)
    assert torch.allclose(q.float(), rot[:, : nh * hs], atol=4e-2, rtol=3e-2)
    assert torch.allclose(kc[:, 5].reshape(B, -1).float(), rot[:, nh * hs:], atol=4e-2, rtol=3e-2)
    assert torch.allclose(vc[:, 5].reshape(B, -1).float(), full[:, (nh + nkv) * hs:], atol=4e-2, rtol=3e-2)
    assert rel_err(q, rot[:, : nh * hs]) < 1e-2 and rel_err(kc[:, 5].reshape(B, -1), rot[:, nh * hs:]) < 1e-2


def test_sampler_step_form_records_the_token_and_advances_the_counters():
    """The decode-step form of the sampler: tokens into the next input and the history column, device counters
    advanced once per launch by the last workgroup, top-1 equal to argmax, draws that change with the position."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(0)
    B, V, steps = 6, 50304, 5
    logits = torch.randn(B, V, device="cuda").bfloat16()
    seed = torch.tensor([1234567], dtype=torch.int64, device="cuda")
    tok = torch.zeros(B, 1, dtype=torch.int64, device="cuda")
    hist = torch.full((B, steps), -1, dtype=torch.int64, device="cuda")
    hpos = torch.tensor([1], dtype=torch.int64, device="cuda")
    pos = torch.tensor([17], dtype=torch.int64, device="cuda")
    fill = torch.tensor([18], dtype=torch.int32, device="cuda")
    ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert ops.sample_step_supported(logits)
    ops.sample_step(logits, 1.0, 1, seed, tok, hist, hpos, pos, fill, ticket)
    assert torch.equal(tok[:, 0], logits.float().argmax(-1))
    assert torch.equal(hist[:, 1], tok[:, 0]) and int((hist >= 0).sum()) == B
    assert (hpos.item(), pos.item(), fill.item(), ticket.item()) == (2, 18, 19, 0)
    draws = []
    for _ in range(3):                                   # same logits, same seed: the position changes the draw
        ops.sample_step(logits, 1.0, 0, seed, tok, hist, hpos, pos, fill, ticket)
        draws.append(tok.clone())
    assert (hpos.item(), pos.item(), fill.item(), ticket.item()) == (5, 21, 22, 0)
    assert torch.equal(hist[:, 4], draws[2][:, 0])
    assert not torch.equal(draws[0], draws[1]) and not torch.equal(draws[1], draws[2])


@pytest.mark.parametrize("act,U", [("gelu", 512), ("swiglu", 512), ("relu", 384)])
def test_grouped_experts_with_dropout_matches_eager(act, U):
    """Dropout on the experts' outputs inside the grouped path (both kernel families): with the mask pinned, output and
    every gradient equal the dense fp32 statement of the layer."""
    from distributed_pytorch_amd import ops
    torch.manual_seed(31)
    N, C, E, k, p = 1536, 256, 4, 2, 0.1
    x = (torch.randn(N, C, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_()
    W = 2 * U if act == "swiglu" else U
    fcs = [(torch.randn(W, C, device=DEV) * 0.05).to(torch.bfloat16).requires_grad_() for _ in range(E)]
    projs = [(torch.randn(C, U, device=DEV) * 0.05).to(torch.bfloat16).requires_grad_() for _ in range(E)]
    logits = torch.randn(N, E, device=DEV)
    idx = logits.topk(k, dim=1).indices
    gates = torch.softmax(logits.gather(1, idx), dim=1).to(torch.bfloat16).requires_grad_()
    keep = torch.rand(N, k, C, device=DEV) >= p
    assert ops.grouped_supported(x, fcs, projs, act)
    out = ops.grouped_experts(x, idx, gates, fcs, projs, act, p_drop=p, keep_pairs=keep)
    g = torch.randn_like(out)
    out.backward(g)
    xr = x.detach().float().requires_grad_()
    gr = gates.detach().float().requires_grad_()
    fr = [w.detach().float().requires_grad_() for w in fcs]
    pr = [w.detach().float().requires_grad_() for w in projs]
    want = ops.grouped_experts_eager(xr, idx, gr, fr, pr, act, p_drop=p, keep_pairs=keep)
    want.backward(g.float())
    assert rel_err(out, want) < 2e-2
    assert rel_err(x.grad, xr.grad) < 3e-2 and rel_err(gates.grad, gr.grad) < 3e-2
    for a, b in zip(fcs + projs, fr + pr):
        assert rel_err(a.grad, b.grad) < 3e-2
    # the random mask drops about p of the outputs and rescales the rest
    o1 = ops.grouped_experts(x.detach(), idx, gates.detach(), [w.detach() for w in fcs], [w.detach() for w in projs], act,
                             p_drop=0.5)
    o0 = ops.grouped_experts(x.detach(), idx, gates.detach(), [w.detach() for w in fcs], [w.detach() for w in projs], act)
    assert not torch.equal(o1, o0) and abs(o1.float().mean().item() - o0.float().mean().item()) < 0.05 * o0.float().abs().mean().item() + 1e-3


@pytest.mark.gpu
def test_transposes_of_resident_weights_refresh_in_one_launch_per_shape():
    """After an optimizer step (the epoch bump) the first stale request refreshes the cached transposes of every
    weight marked resident: same buffers, new contents; weights not marked resident keep the one-by-one path."""
    import importlib
    L = importlib.import_module("distributed_pytorch_amd.ops.linear")   # (the package exports a function of that name)
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(256, 128, device="cuda", dtype=torch.bfloat16)) for _ in range(3)]
    ws += [torch.nn.Parameter(torch.randn(64, 512, device="cuda", dtype=torch.bfloat16)) for _ in range(2)]
    other = torch.nn.Parameter(torch.randn(256, 128, device="cuda", dtype=torch.bfloat16))
    for w in ws:
        w._dpa_resident = True
    first = [L.transposed_weight(w) for w in ws + [other]]
    for w, t in zip(ws + [other], first):
        assert torch.equal(t, w.detach().t().contiguous())
    with torch.no_grad():
        for w in ws + [other]:
            w.data.view(-1)[:].mul_(-0.5)          # an update that autograd's version counter may or may not see
    L.bump_weights_epoch()
    t0 = L.transposed_weight(ws[0])                # refreshes all five resident entries
    for w, old in zip(ws, first):
        ent = L._WT_CACHE[id(w)]
        assert ent[0][0] == L._WEIGHTS_EPOCH and ent[1].data_ptr() == old.data_ptr()
        assert torch.equal(ent[1], w.detach().t().contiguous())
    assert t0.data_ptr() == first[0].data_ptr()
    assert L._WT_CACHE[id(other)][0][0] != L._WEIGHTS_EPOCH      # not resident: still stale until asked for
    assert torch.equal(L.transposed_weight(other), other.detach().t().contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("nslabs", [1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 23, 28, 64])
@pytest.mark.parametrize("out_dtype,accumulate", [(torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, True)])
def test_slab_fold_every_slab_count(nslabs, out_dtype, accumulate):
    """The fold of the weight-gradient slabs asks for eight, then four, then up to three slabs at a time: every
    count from one slab to the largest split hits a different mix of those branches."""
    torch.manual_seed(nslabs)
    N, K = 96, 200
    ws = torch.randn(nslabs, N, K, device=DEV, dtype=torch.float32)
    out = torch.randn(N, K, device=DEV, dtype=torch.float32).to(out_dtype)
    before = out.float().clone()
    _ops().ws_to(ws, out, accumulate)
    ref = ws.double().sum(0) + (before.double() if accumulate else 0)
    tol = 1e-5 if out_dtype == torch.float32 else 5e-3
    assert rel_err(out.float(), ref.float()) < tol


@pytest.mark.gpu
def test_layernorm_streaming_policy_instantiation():
    """Activations of 256 MiB and more go through the LayerNorm kernels built with the non-temporal cache policy; the
    threshold is read once per process, so a child process with DPA_LN_STREAM_MB=0 runs the small-shape checks on
    those instantiations (plain, with the residual add, backward with the residual gradient)."""
    import os
    import subprocess
    import sys
    code = r'''
import torch, torch.nn.functional as F
from distributed_pytorch_amd import ops
torch.manual_seed(0)
for rows, C in ((300, 768), (129, 1024), (64, 2048)):
    x = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(rows, C, device="cuda", dtype=torch.bfloat16)
    w = (1 + 0.1 * torch.randn(C, device="cuda")).to(torch.bfloat16).requires_grad_()
    b = (0.1 * torch.randn(C, device="cuda")).to(torch.bfloat16).requires_grad_()
    xs = x.clone().requires_grad_(); rs = r.clone().requires_grad_()
    s, y = ops.add_layer_norm(xs, rs, w, b, 1e-5)
    (y.float().square().sum() + s.float().sum()).backward()
    xr = x.float().requires_grad_(); rr = r.float().requires_grad_()
    wr = w.detach().float().requires_grad_(); br = b.detach().float().requires_grad_()
    sr = xr + rr
    yr = F.layer_norm(sr.to(torch.bfloat16).float(), (C,), wr, br, 1e-5)
    def rel(a, c): return ((a.float() - c.float()).norm() / (c.float().norm() + 1e-12)).item()
    assert rel(y, yr) < 1e-2, (rows, C, rel(y, yr))
    (yr.square().sum() + sr.sum()).backward()
    assert rel(xs.grad, xr.grad) < 2e-2 and rel(w.grad, wr.grad) < 2e-2 and rel(b.grad, br.grad) < 2e-2, (rows, C)
print("streaming LayerNorm ok")
'''
    env = dict(os.environ, DPA_LN_STREAM_MB="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "streaming LayerNorm ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_transposed_weight_cache_entries_leave_with_their_weight():
    """The cache of transposed weights is keyed by the weight tensor object; an entry whose weight has died is removed
    with it (a weight handed out as a fresh view every step once left one transposed copy behind per step)."""
    import importlib
    lin = importlib.import_module("distributed_pytorch_amd.ops.linear")
    base = torch.randn(256, 128, device=DEV, dtype=torch.bfloat16)
    keep = base.view_as(base)
    wt = lin.transposed_weight(keep)
    assert torch.equal(wt, base.t().contiguous())
    n0 = len(lin._WT_CACHE)
    for _ in range(10):
        v = base.view_as(base)
        assert torch.equal(lin.transposed_weight(v), wt)
        del v
    assert len(lin._WT_CACHE) == n0
    assert lin.transposed_weight(keep) is wt          # the live weight's entry is still served from the cache
