"""causal_conv1d_fn on the MI355X kernels.

Mirrors ``causal_conv1d.causal_conv1d_fn(x, weight, bias=None, activation=None)``
of the causal-conv1d package the reference installs (README.md:55) and its mixer
calls before x_proj.
"""
from __future__ import annotations

import torch

from . import _lib


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in (None, "silu", "swish"):
            raise NotImplementedError("activation must be None, silu, or swish")
        _lib.require_gpu(x, "causal_conv1d_fn")
        lib = _lib.load()
        if x.dim() != 3 or weight.dim() != 2 or weight.shape[0] != x.shape[1]:
            raise ValueError("causal_conv1d_fn: x must be (batch, dim, seqlen), weight (dim, width)")
        batch, dim, L = x.shape
        W = weight.shape[1]
        xc = x if (x.stride(2) == 1 and x.stride(1) == L) else x.contiguous()   # batch-strided views are fine
        wc = weight.float().contiguous()
        bc = None if bias is None else bias.float().contiguous()
        out = torch.empty(batch, dim, L, device=x.device, dtype=x.dtype)
        silu = int(activation is not None)
        with torch.cuda.device(x.device), _lib.timed("conv1d_fwd", x.device):
            rc = lib.simamba_causal_conv1d_fwd(_lib.ptr(xc), _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(out),
                                               batch, dim, L, W, silu, _lib.dtype_code(x.dtype),
                                               xc.stride(0), _lib.stream_ptr(x.device))
        _lib.check(rc, "simamba_causal_conv1d_fwd")
        ctx.silu = silu
        ctx.w_dtype = weight.dtype
        ctx.b_dtype = None if bias is None else bias.dtype
        ctx.save_for_backward(xc, wc, bc)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, wc, bc = ctx.saved_tensors
        lib = _lib.load()
        batch, dim, L = xc.shape
        W = wc.shape[1]
        dout = dout.to(xc.dtype).contiguous()
        dx = torch.empty(batch, dim, L, device=xc.device, dtype=xc.dtype)
        dw = torch.empty_like(wc)
        db = torch.empty(dim, device=xc.device, dtype=torch.float32) if bc is not None else None
        with torch.cuda.device(xc.device), _lib.timed("conv1d_bwd", xc.device):
            rc = lib.simamba_causal_conv1d_bwd(_lib.ptr(xc), _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(dout),
                                               _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db),
                                               batch, dim, L, W, ctx.silu, _lib.dtype_code(xc.dtype),
                                               xc.stride(0), 0, _lib.stream_ptr(xc.device))
        _lib.check(rc, "simamba_causal_conv1d_bwd")
        return dx, dw.to(ctx.w_dtype), None if db is None else db.to(ctx.b_dtype), None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width); bias: (dim); activation in {None, silu, swish}."""
    return CausalConv1dFn.apply(x, weight, bias, activation)
