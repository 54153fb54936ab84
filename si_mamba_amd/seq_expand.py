"""The first block's per-token work on the G distinct tokens instead of the 2 k G sequence positions.

The reference builds the block stack's input by gathering the SAME G patch tokens (and their position embeddings)
in 2 k orders (models/point_mamba.py:889-898, :982-989) and then runs Block 0 on all L = 2 k G positions
(models/block.py:56-72).  Add, LayerNorm and the mixer's in_proj act on each token by itself, so

    in_proj(LN(gather(tokens + pos)))  ==  gather(in_proj(LN(tokens + pos)))

exactly: the left side is what the reference computes, the right side costs 1/(2k) of the in_proj flops (forward,
input gradient and weight gradient alike) plus one copy kernel each way (csrc/seq_gather.hip).  Valid whenever every
position of the sequence is one of the G tokens and nothing position-dependent (dropout with p > 0) sits between the
gather and the first block; ``MixerModel.forward(..., token_index=...)`` takes this route, everything else the
reference's.
"""
from __future__ import annotations

import torch

from . import _lib


def inverse_positions(token_index, G):
    """(B, L) token of every sequence position -> (B, G, R) int32 positions of every token.  PRECONDITION (the caller's
    to state, MixerModel.forward(balanced_index=True)): every token occurs exactly R = L / G times in every row -- a
    concatenation of permutations.  With unequal counts the grouping below would put positions into the wrong tokens'
    rows and the adjoint kernel would sum the wrong gradients; checking it here would cost a device synchronisation
    per forward, so it is a contract, enforced in the tests (tests/test_gpu_model.py).  None when L % G != 0."""
    B, L = token_index.shape
    if L % G:
        return None
    order = torch.sort(token_index, dim=1, stable=True)[1]              # positions grouped by token, ascending
    return order.view(B, G, L // G).to(torch.int32).contiguous()


class SeqGatherLastFn(torch.autograd.Function):
    """out[b, c, l] = x[b, c, idx[b, l]] for x (B, C, G) -> (B, C, L)."""

    @staticmethod
    def forward(ctx, x, idx32, inv32):
        _lib.require_gpu(x, "seq_gather_last")
        lib = _lib.load()
        xc = x.contiguous()
        B, C, G = xc.shape
        L = idx32.shape[1]
        out = torch.empty(B, C, L, device=xc.device, dtype=xc.dtype)
        with torch.cuda.device(xc.device), _lib.timed("seq_gather_fwd", xc.device):
            rc = lib.simamba_seq_gather_fwd(xc.data_ptr(), idx32.data_ptr(), out.data_ptr(), B, C, G, L, 0,
                                            _lib.dtype_code(xc.dtype), _lib.stream_ptr(xc.device))
        _lib.check(rc, "simamba_seq_gather_fwd")
        ctx.save_for_backward(inv32)
        ctx.meta = (B, C, G, L)
        return out

    @staticmethod
    def backward(ctx, dout):
        (inv32,) = ctx.saved_tensors
        B, C, G, L = ctx.meta
        lib = _lib.load()
        d = dout
        if d.stride(2) != 1 or d.stride(1) != L or d.stride(0) % 4:
            d = d.contiguous()
        din = torch.empty(B, C, G, device=d.device, dtype=d.dtype)
        with torch.cuda.device(d.device), _lib.timed("seq_gather_bwd", d.device):
            rc = lib.simamba_seq_gather_bwd(d.data_ptr(), inv32.data_ptr(), din.data_ptr(), B, C, G, L, L // G,
                                            d.stride(0), _lib.dtype_code(d.dtype), _lib.stream_ptr(d.device))
        _lib.check(rc, "simamba_seq_gather_bwd")
        return din, None, None


def seq_gather_last(x, idx32, inv32):
    return SeqGatherLastFn.apply(x, idx32, inv32)


def expansion_ok(x, token_index):
    """Shapes the copy kernels take (include/simamba.h)."""
    G, L = x.shape[1], token_index.shape[1]
    return (x.is_cuda and x.dim() == 3 and G % 4 == 0 and L % 4 == 0 and G <= 256 and L <= 2048 and L % G == 0
            and L // G <= 8)
