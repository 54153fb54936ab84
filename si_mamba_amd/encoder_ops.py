"""Streaming ops of the patch encoder on the MI355X kernels (csrc/bn_relu.hip).

  bn_relu_fn      relu(BatchNorm1d(x + g))   reference models/point_mamba.py:46-49, :52-55
                  (nn.BatchNorm1d + nn.ReLU after a 1x1 Conv1d, token-major here); ``g`` is the per-patch
                  additive term that replaces the reference's concat of the global feature (:64-66)
  group_max_fn    max over the points of a patch   reference :63, :68  (torch.max(..., dim=2)[0])

Both are plain autograd Functions over the C ABI; BatchNorm keeps nn.BatchNorm1d's buffers (running_mean,
running_var, num_batches_tracked) and train/eval semantics.
"""
from __future__ import annotations

import torch

from . import _lib


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
        part = torch.empty(lib.simamba_bn_relu_grid(rows), 2, C, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _lib.timed("bn_relu_fwd", dev):
            rc = lib.simamba_bn_relu_fwd(xc.data_ptr(), _lib.ptr(g), int(group), _lib.ptr(w), _lib.ptr(b),
                                         _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), float(eps),
                                         int(bool(training)), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                         part.data_ptr(), rows, C, code, _lib.stream_ptr(dev))
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
        dg = None if g is None else torch.empty_like(g)
        dw = torch.empty(C, device=dev, dtype=torch.float32)
        db = torch.empty(C, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev), _lib.timed("bn_relu_bwd", dev):
            rc = lib.simamba_bn_relu_bwd(dyc.data_ptr(), xc.data_ptr(), _lib.ptr(g), group, _lib.ptr(w), _lib.ptr(b),
                                         mean.data_ptr(), invstd.data_ptr(), dx.data_ptr(), _lib.ptr(dg),
                                         dw.data_ptr(), db.data_ptr(), ctx.part.data_ptr(), rows, C, code,
                                         int(training), _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_bn_relu_bwd")
        return (dx.to(xdtype), None if dg is None else dg.to(gdtype), None,
                None if wdtype is None else dw.to(wdtype), None if bdtype is None else db.to(bdtype),
                None, None, None, None, None)


def bn_relu_fn(x, bn: torch.nn.BatchNorm1d, gterm=None, group=0):
    """relu(bn(x + gterm[row // group])) for token-major x (rows, C) with ``bn``'s parameters, buffers and mode."""
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
