"""Streaming ops of the patch encoder on the MI355X kernels (csrc/bn_relu.hip).

  bn_relu_fn      relu(BatchNorm1d(x + g))   reference models/point_mamba.py:46-49, :52-55
                  (nn.BatchNorm1d + nn.ReLU after a 1x1 Conv1d, token-major here); ``g`` is the per-patch
                  additive term that replaces the reference's concat of the global feature (:64-66)
  group_max_fn    max over the points of a patch   reference :63, :68  (torch.max(..., dim=2)[0])

  token_linear    y = x W^T + b on a token-major (rows, C) view (the reference's 1x1 Conv1d, :44-57) whose weight
                  gradient is a split-K batched product: dW = sum_s dY_s^T X_s over 64 row slabs, fp32 partials.
                  The one-GEMM form autograd would run has K = rows = 262 144 and a (C_out, C_in) result of a few
                  tiles: the library's kernels for it leave most of the chip idle (bf16: 540-790 us per layer
                  against 44-143 us split; fp32: 443-1 778 against 156-750 us; tools/bench_wgrad_splitk.py).

bn_relu_fn and group_max_fn are plain autograd Functions over the C ABI; BatchNorm keeps nn.BatchNorm1d's buffers (running_mean,
running_var, num_batches_tracked) and train/eval semantics.
"""
from __future__ import annotations

import torch

from . import _lib


_BN_MAX_C = 1024     # channels per kernel call; wider layers go slice by slice (in place, row stride = C)
_BN_CHUNK = 256      # rows per workgroup in csrc/bn_relu.hip: the granularity of its per-group dx sums


_SPLITK = 64         # row slabs of token_linear's weight gradient
_SPLITK_MIN_ROWS = 512   # per slab; below it the plain product is used


class TokenLinearFn(torch.autograd.Function):
    """F.linear on (rows, C_in) tokens with a split-K weight gradient (library GEMMs; no custom kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch.nn.functional.linear(x, weight, bias)       # under autocast: the reference's rounding (bf16 GEMM)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        io = dy.dtype
        rows, n_out = dy.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (dy @ weight.to(io)).to(x.dtype)
        if ctx.needs_input_grad[1]:
            xc = x.to(io).contiguous()
            if rows % _SPLITK == 0 and rows // _SPLITK >= _SPLITK_MIN_ROWS:
                m = rows // _SPLITK
                a = dy.view(_SPLITK, m, n_out).transpose(1, 2)
                b = xc.view(_SPLITK, m, xc.shape[1])
                part = torch.bmm(a, b) if io == torch.float32 else torch.bmm(a, b, out_dtype=torch.float32)
                dw = part.sum(0).to(weight.dtype)
            else:
                dw = (dy.t() @ xc).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0, dtype=torch.float32).to(weight.dtype)
        return dx, dw, db


def token_linear(x, weight, bias=None):
    """(rows, C_in) @ (C_out, C_in)^T + bias.  GPU tensors take the split-K weight gradient; anything else is
    torch.nn.functional.linear."""
    if x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return TokenLinearFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def _slices(C):
    return [(c0, min(c0 + _BN_MAX_C, C)) for c0 in range(0, C, _BN_MAX_C)]


def _off(t, elems):
    """data_ptr of a contiguous tensor advanced by `elems` elements (None stays None)."""
    return None if t is None else t.data_ptr() + elems * t.element_size()


class BnReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gterm, group, weight, bias, running_mean, running_var, momentum, eps, training):
        _lib.require_gpu(x, "bn_relu_fn")
        lib = _lib.load()
        xc = x.contiguous()
        rows, C = xc.shape
        dev = xc.device
        code = _lib.dtype_code(xc.dtype)
        g = None if gterm is None else gterm.float().contiguous()
        w = None if weight is None else weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        y = torch.empty_like(xc)
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        invstd = torch.empty(C, device=dev, dtype=torch.float32)
        grid = lib.simamba_bn_relu_grid(rows)
        part = torch.empty(grid, 2, min(C, _BN_MAX_C), device=dev, dtype=torch.float32)
        gs = []
        with torch.cuda.device(dev), _lib.timed("bn_relu_fwd", dev):
            for c0, c1 in _slices(C):
                gsl = None if g is None else (g if (c0 == 0 and c1 == C) else g[:, c0:c1].contiguous())
                gs.append(gsl)
                rc = lib.simamba_bn_relu_fwd(_off(xc, c0), _lib.ptr(gsl), int(group), _off(w, c0), _off(b, c0),
                                             _off(running_mean, c0), _off(running_var, c0), float(momentum),
                                             float(eps), int(bool(training)), _off(y, c0), _off(mean, c0),
                                             _off(invstd, c0), part.data_ptr(), rows, c1 - c0, C, code,
                                             _lib.stream_ptr(dev))
                _lib.check(rc, "simamba_bn_relu_fwd")
        ctx.save_for_backward(xc, g, w, b, mean, invstd)
        ctx.meta = (int(group), bool(training), code, x.dtype,
                    None if gterm is None else gterm.dtype,
                    None if weight is None else weight.dtype, None if bias is None else bias.dtype)
        ctx.part = part
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, g, w, b, mean, invstd = ctx.saved_tensors
        group, training, code, xdtype, gdtype, wdtype, bdtype = ctx.meta
        lib = _lib.load()
        rows, C = xc.shape
        dev = xc.device
        dyc = dy.to(xc.dtype).contiguous()
        dx = torch.empty_like(xc)
        dw = torch.empty(C, device=dev, dtype=torch.float32)
        db = torch.empty(C, device=dev, dtype=torch.float32)
        # the kernel sums dx over runs of dgroup <= 256 rows; wider groups are finished here
        dgroup = 0 if g is None else (group if _BN_CHUNK % group == 0 else _BN_CHUNK)
        if g is not None and dgroup == _BN_CHUNK and group % _BN_CHUNK != 0:
            raise ValueError("bn_relu_fn: group must divide 256 or be a multiple of it")
        dgs = []
        with torch.cuda.device(dev), _lib.timed("bn_relu_bwd", dev):
            for c0, c1 in _slices(C):
                gsl = None if g is None else (g if (c0 == 0 and c1 == C) else g[:, c0:c1].contiguous())
                dg = None if g is None else torch.empty(rows // dgroup, c1 - c0, device=dev, dtype=torch.float32)
                dgs.append(dg)
                rc = lib.simamba_bn_relu_bwd(_off(dyc, c0), _off(xc, c0), _lib.ptr(gsl), group, _off(w, c0),
                                             _off(b, c0), _off(mean, c0), _off(invstd, c0), _off(dx, c0), _lib.ptr(dg),
                                             dgroup, _off(dw, c0), _off(db, c0), ctx.part.data_ptr(), rows, c1 - c0, C,
                                             code, int(training), _lib.stream_ptr(dev))
                _lib.check(rc, "simamba_bn_relu_bwd")
        dgt = None
        if g is not None:
            dgt = dgs[0] if len(dgs) == 1 else torch.cat(dgs, dim=1)
            if dgroup != group:
                dgt = dgt.view(rows // group, group // dgroup, C).sum(1)
            dgt = dgt.to(gdtype)
        return (dx.to(xdtype), dgt, None,
                None if wdtype is None else dw.to(wdtype), None if bdtype is None else db.to(bdtype),
                None, None, None, None, None)


def bn_relu_fn(x, bn: torch.nn.BatchNorm1d, gterm=None, group=0):
    """relu(bn(x + gterm[row // group])) for token-major x (rows, C) with ``bn``'s parameters, buffers and mode.

    The HIP kernel computes PER-RANK batch statistics, which is what ``nn.BatchNorm1d`` means.  Any other module
    type -- in particular the ``nn.SyncBatchNorm`` that ``--sync_bn`` puts in its place
    (reference tools/runner_finetune.py:121-122, runner_pretrain.py:111-112), whose training statistics are reduced
    across ranks -- goes through the module itself (stock torch kernels on the same device), so its semantics are
    kept instead of being silently replaced."""
    if type(bn) is not torch.nn.BatchNorm1d:
        if gterm is not None:
            x = x + gterm.to(x.dtype).repeat_interleave(group, dim=0)
        return torch.relu(bn(x))
    training = bn.training or bn.running_mean is None
    momentum = bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if momentum is None:                       # cumulative moving average (nn.BatchNorm1d semantics)
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm = bn.running_mean if (bn.track_running_stats and (bn.training or not training)) else None
    rv = bn.running_var if rm is not None else None
    return BnReluFn.apply(x, gterm, group, bn.weight, bn.bias, rm, rv, 0.0 if momentum is None else momentum,
                          bn.eps, training)


class GroupMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_gpu(x, "group_max_fn")
        lib = _lib.load()
        xc = x.contiguous()
        groups, n, C = xc.shape
        dev = xc.device
        code = _lib.dtype_code(xc.dtype)
        out = torch.empty(groups, C, device=dev, dtype=xc.dtype)
        idx = torch.empty(groups, C, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            rc = lib.simamba_group_max_fwd(xc.data_ptr(), out.data_ptr(), idx.data_ptr(), groups, n, C, code,
                                           _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_group_max_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (groups, n, C, code, xc.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        groups, n, C, code, dtype = ctx.meta
        lib = _lib.load()
        d = dout.to(dtype).contiguous()
        dx = torch.empty(groups, n, C, device=d.device, dtype=dtype)
        with torch.cuda.device(d.device):
            rc = lib.simamba_group_max_bwd(d.data_ptr(), idx.data_ptr(), dx.data_ptr(), groups, n, C, code,
                                           _lib.stream_ptr(d.device))
        _lib.check(rc, "simamba_group_max_bwd")
        return dx


def group_max_fn(x):
    """(groups, n, C) -> (groups, C): max over the n points of each patch."""
    return GroupMaxFn.apply(x)
