"""out_proj -> (+ DropPath-scaled residual) -> LayerNorm as one op on the matrix cores (csrc/out_norm_bf16.hip).

What the reference computes across a block boundary under autocast -- the mixer's ``out_proj`` (upstream
``mamba_inner_fn``, reached from models/block.py:72), then ``residual = drop_path(hidden) + residual`` and
``hidden = norm(residual)`` at the top of the NEXT block (models/block.py:56-58) or the stack's final norm
(models/point_mamba.py:257-258):

    hidden   = y^T @ out_proj.weight^T          (bf16 GEMM output)
    residual = hidden * rowscale + residual     (fp32)
    normed   = LayerNorm(residual)

One kernel forward: the out_proj result never goes to memory.  Backward: the LayerNorm / add backward kernel of
add_norm.py, then out_proj's input gradient through the hand-written token-times-weight kernel (csrc/in_proj_bf16.hip;
the library GEMM where its grid would not fill the chip) and its weight gradient through the library.  bf16 operands only (the autocast
configurations); MixerModel.forward takes this route when it applies and the reference's op-by-op route otherwise.
"""
from __future__ import annotations

import torch

from . import _lib
from .mamba_inner import _sum_bmm, tokens_times_weight


def out_proj_add_ln_ok(y, out_w, d_model):
    """Shapes the kernel takes (include/simamba.h)."""
    B, D, L = y.shape
    return (y.is_cuda and y.dtype == torch.bfloat16 and y.stride(2) == 1 and y.stride(1) == L and y.stride(0) == D * L
            and d_model % 128 == 0 and d_model <= 384 and D % 64 == 0 and L % 8 == 0 and D * L * 2 < 2 ** 32 - 65536
            and tuple(out_w.shape) == (d_model, D) and y.data_ptr() % 16 == 0)


class OutProjAddLnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, out_w, residual, ln_w, ln_b, eps, rowscale, out_dtype):
        _lib.require_gpu(y, "out_proj_add_ln_fn")
        lib = _lib.load()
        Bsz, D, L = y.shape
        C = out_w.shape[0]
        dev = y.device
        wc = out_w.to(torch.bfloat16).contiguous()
        res = None if residual is None else residual.float().contiguous()
        lw = ln_w.float().contiguous()
        lb = None if ln_b is None else ln_b.float().contiguous()
        rs = None if (rowscale is None or res is None) else rowscale.float().contiguous()
        res_out = torch.empty(Bsz, L, C, device=dev, dtype=torch.float32)
        normed = torch.empty(Bsz, L, C, device=dev, dtype=out_dtype)
        mean = torch.empty(Bsz * L, device=dev, dtype=torch.float32)
        rstd = torch.empty(Bsz * L, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _lib.timed("out_proj_add_ln_fwd", dev):
            rc = lib.simamba_out_proj_add_ln_fwd(y.data_ptr(), wc.data_ptr(), _lib.ptr(res), _lib.ptr(rs), lw.data_ptr(),
                                                 _lib.ptr(lb), res_out.data_ptr(), normed.data_ptr(), mean.data_ptr(),
                                                 rstd.data_ptr(), Bsz, D, L, C, float(eps), _lib.dtype_code(out_dtype),
                                                 _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_out_proj_add_ln_fwd")
        ctx.save_for_backward(y, wc, res_out, mean, rstd, lw, rs)
        ctx.meta = (Bsz, L, C, _lib.dtype_code(out_dtype), residual is not None,
                    None if residual is None else residual.dtype, ln_w.dtype, ln_b is not None, out_w.dtype)
        return normed, res_out

    @staticmethod
    def backward(ctx, dnormed, dres_out):
        y, wc, res_out, mean, rstd, lw, rs = ctx.saved_tensors
        Bsz, L, C, ocode, has_res, res_dtype, lwdtype, has_bias, owdtype = ctx.meta
        lib = _lib.load()
        dev = y.device
        dn = dnormed.contiguous()
        dro = None if dres_out is None else dres_out.float().contiguous()
        dres = torch.empty(res_out.shape, device=dev, dtype=torch.float32) if has_res else None
        dhid = torch.empty(res_out.shape, device=dev, dtype=torch.bfloat16)     # gradient of the bf16 out_proj output
        grid = lib.simamba_add_layer_norm_grid(Bsz, L)
        part = torch.empty(grid, 2, C, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _lib.timed("add_ln_bwd", dev):
            rc = lib.simamba_add_layer_norm_bwd(dn.data_ptr(), _lib.ptr(dro), res_out.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), lw.data_ptr(), _lib.ptr(rs), _lib.ptr(dres),
                                                dhid.data_ptr(), part.data_ptr(), Bsz, L, C, _lib.BF16, ocode,
                                                _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_add_layer_norm_bwd")
        dwb = part.sum(0)
        dy = tokens_times_weight(dhid, wc.t())                                    # (B, D, L)
        d_out_w = _sum_bmm(dhid.transpose(1, 2), y.transpose(1, 2))               # (C, D)
        return (dy, d_out_w.to(owdtype), None if not has_res else dres.to(res_dtype), dwb[0].to(lwdtype),
                dwb[1].to(lwdtype) if has_bias else None, None, None, None)


def out_proj_add_ln_fn(y, out_w, residual, ln_w, ln_b, eps=1e-5, rowscale=None, out_dtype=torch.bfloat16):
    """y (B, D, L) bf16 -> (normed (B, L, C) out_dtype, residual_out (B, L, C) fp32)."""
    return OutProjAddLnFn.apply(y, out_w, residual, ln_w, ln_b, eps, rowscale, out_dtype)
