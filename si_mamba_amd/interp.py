"""3-nearest-centre inverse-distance feature interpolation on the MI355X kernels (csrc/interp.hip): the
arithmetic of the reference's PointNetFeaturePropagation.forward
(part_segmentation/models/pointnet2_utils.py:262-305) between its ``square_distance`` and its MLP."""
from __future__ import annotations

import torch

from . import _lib


def three_nn(xyz1, xyz2):
    """xyz1 (B,N,3) query points, xyz2 (B,S,3) centres -> idx (B,N,3) int32, weight (B,N,3) fp32."""
    _lib.require_gpu(xyz1, "three_nn")
    lib = _lib.load()
    a = xyz1.detach().float().contiguous()
    c = xyz2.detach().float().contiguous()
    B, N, _ = a.shape
    S = c.shape[1]
    idx = torch.empty(B, N, 3, device=a.device, dtype=torch.int32)
    w = torch.empty(B, N, 3, device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        rc = lib.simamba_three_nn(a.data_ptr(), c.data_ptr(), idx.data_ptr(), w.data_ptr(), B, N, S,
                                  _lib.stream_ptr(a.device))
    _lib.check(rc, "simamba_three_nn")
    return idx, w


class ThreeInterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, weight):
        _lib.require_gpu(feats, "three_interpolate")
        lib = _lib.load()
        f = feats.contiguous()
        B, S, C = f.shape
        N = idx.shape[1]
        code = _lib.dtype_code(f.dtype)
        out = torch.empty(B, N, C, device=f.device, dtype=f.dtype)
        with torch.cuda.device(f.device), _lib.timed("interp_fwd", f.device):
            rc = lib.simamba_three_interpolate_fwd(f.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                                   B, N, S, C, code, _lib.stream_ptr(f.device))
        _lib.check(rc, "simamba_three_interpolate_fwd")
        ctx.save_for_backward(idx, weight)
        ctx.meta = (B, N, S, C, code, feats.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, weight = ctx.saved_tensors
        B, N, S, C, code, dtype = ctx.meta
        lib = _lib.load()
        d = dout.to(dtype).contiguous()
        df = torch.empty(B, S, C, device=d.device, dtype=torch.float32)
        with torch.cuda.device(d.device), _lib.timed("interp_bwd", d.device):
            rc = lib.simamba_three_interpolate_bwd(d.data_ptr(), idx.data_ptr(), weight.data_ptr(), df.data_ptr(),
                                                   B, N, S, C, code, _lib.stream_ptr(d.device))
        _lib.check(rc, "simamba_three_interpolate_bwd")
        return df.to(dtype), None, None


def three_interpolate(feats, idx, weight):
    """feats (B,S,C) -> (B,N,C): sum_k weight[b,n,k] * feats[b, idx[b,n,k], :]."""
    return ThreeInterpolateFn.apply(feats, idx, weight)
