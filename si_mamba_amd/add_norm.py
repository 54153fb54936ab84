"""Fused DropPath-scaled residual add + LayerNorm on the MI355X kernel (csrc/add_norm.hip).

Computes exactly what the reference's Block.forward does before the mixer (models/block.py:56-60):

    residual = drop_path(hidden) + residual        (or hidden when residual is None)
    normed   = LayerNorm(residual)

as one streaming pass forward and one backward.  ``rowscale`` is the per-sample DropPath factor
(keep-mask / keep-prob) or None.
"""
from __future__ import annotations

import torch

from . import _lib


class AddLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, residual, weight, bias, eps, rowscale, out_dtype):
        _lib.require_gpu(hidden, "add_layer_norm_fn")
        lib = _lib.load()
        h = hidden.contiguous()
        Bsz, rows, dim = h.shape[0], h[0].numel() // h.shape[-1], h.shape[-1]
        dev = h.device
        hcode = _lib.dtype_code(h.dtype)
        ocode = _lib.dtype_code(out_dtype)
        res = None if residual is None else residual.float().contiguous()
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        rs = None if (rowscale is None or res is None) else rowscale.float().contiguous()
        alias = res is None and h.dtype == torch.float32        # residual_out is hidden itself
        res_out = h if alias else torch.empty(h.shape, device=dev, dtype=torch.float32)
        normed = torch.empty(h.shape, device=dev, dtype=out_dtype)
        mean = torch.empty(Bsz * rows, device=dev, dtype=torch.float32)
        rstd = torch.empty(Bsz * rows, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _lib.timed("add_ln_fwd", dev):
            rc = lib.simamba_add_layer_norm_fwd(h.data_ptr(), _lib.ptr(res), _lib.ptr(rs), w.data_ptr(), _lib.ptr(b),
                                                None if alias else res_out.data_ptr(), normed.data_ptr(),
                                                mean.data_ptr(), rstd.data_ptr(), Bsz, rows, dim, float(eps),
                                                hcode, ocode, _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_add_layer_norm_fwd")
        ctx.save_for_backward(res_out, mean, rstd, w, rs)
        ctx.meta = (Bsz, rows, dim, hcode, ocode, h.dtype, residual is not None,
                    None if residual is None else residual.dtype, weight.dtype, bias is not None)
        return normed, res_out

    @staticmethod
    def backward(ctx, dnormed, dres_out):
        res_out, mean, rstd, w, rs = ctx.saved_tensors
        Bsz, rows, dim, hcode, ocode, hdtype, has_res, res_dtype, wdtype, has_bias = ctx.meta
        lib = _lib.load()
        dev = res_out.device
        dn = dnormed.contiguous()
        dro = None if dres_out is None else dres_out.float().contiguous()
        # gradient w.r.t. residual (fp32) and w.r.t. hidden; they are the same tensor unless DropPath rescales
        # hidden or hidden is not fp32
        need_split = (rs is not None) or (hdtype != torch.float32) or not has_res
        dres = torch.empty(res_out.shape, device=dev, dtype=torch.float32) if has_res else None
        dhid = torch.empty(res_out.shape, device=dev, dtype=hdtype) if need_split else None
        grid = lib.simamba_add_layer_norm_grid(Bsz, rows)
        part = torch.empty(grid, 2, dim, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _lib.timed("add_ln_bwd", dev):
            rc = lib.simamba_add_layer_norm_bwd(dn.data_ptr(), _lib.ptr(dro), res_out.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), w.data_ptr(), _lib.ptr(rs), _lib.ptr(dres),
                                                _lib.ptr(dhid), part.data_ptr(), Bsz, rows, dim, hcode, ocode,
                                                _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_add_layer_norm_bwd")
        dwb = part.sum(0)
        if dhid is None:
            dhid = dres
        return (dhid, None if not has_res else dres.to(res_dtype), dwb[0].to(wdtype),
                dwb[1].to(wdtype) if has_bias else None, None, None, None)


def add_layer_norm_fn(hidden, residual, weight, bias, eps=1e-5, rowscale=None, out_dtype=None):
    """-> (normed, residual_out).  hidden: (B, ..., dim); residual: same shape or None; rowscale: (B,) or None."""
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else weight.dtype
    return AddLayerNormFn.apply(hidden, residual, weight, bias, eps, rowscale, out_dtype)
