"""Token + position embedding as one op.

Forward: ``out[b, t] = wte[idx[b, t]] + pos[pos0 + t]`` (``pos`` optional).
Backward: the token-table gradient is a sorted, atomic-free segment sum added into the
table's gradient rows; the position-table gradient is a batch reduction.

Reference behaviour: ``nn.Embedding`` lookups added together
(reference: single-gpu/model.py:652-666).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .._ext import ops, use_hip


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, wte, pos, pos0, pos_requires_grad):
        idx = idx.contiguous()
        out = ops().embed_fwd(idx, wte, pos, pos0)
        ctx.save_for_backward(idx)
        ctx.meta = (wte.shape, wte.dtype, None if pos is None else pos.shape, pos0, pos_requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        wshape, wdtype, pshape, pos0, pos_rg = ctx.meta
        dout = dout.contiguous()
        dwte = None
        if ctx.needs_input_grad[1]:
            dwte = torch.zeros(wshape, dtype=wdtype, device=dout.device)
            ids_sorted, perm = torch.sort(idx.view(-1))
            ops().embed_bwd(dout.view(-1, wshape[1]), ids_sorted, perm, dwte)
        dpos = None
        if pshape is not None and pos_rg and ctx.needs_input_grad[2]:
            dpos = torch.zeros(pshape, dtype=wdtype, device=dout.device)
            ops().pos_bwd(dout, dpos, pos0)
        return None, dwte, dpos, None, None


def embedding(idx: torch.Tensor, wte: torch.Tensor, pos: torch.Tensor | None = None, pos0: int = 0) -> torch.Tensor:
    T = idx.shape[1]
    if (use_hip(idx, wte) and wte.dtype in (torch.bfloat16, torch.float32) and wte.shape[1] % 8 == 0
            and (pos is None or (pos.is_cuda and pos.dtype == wte.dtype))):
        return _EmbedFn.apply(idx, wte, pos, int(pos0), bool(pos is not None and pos.requires_grad))
    x = F.embedding(idx, wte)
    if pos is not None:
        x = x + pos[pos0:pos0 + T].to(x.dtype)
    return x
