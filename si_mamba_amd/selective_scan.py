"""selective_scan_fn on the MI355X kernels.

Mirrors the operator the reference's mixer uses
(``mamba_ssm.ops.selective_scan_interface.selective_scan_fn``; reached from
reference models/block.py:72 through the Mamba built at models/point_mamba.py:162):
same argument names, meaning and return values.
"""
from __future__ import annotations

import torch

from . import _lib


def _rows_contiguous(t):
    """(B, D, L) view whose (D, L) planes are dense: only the batch stride may differ."""
    return t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2]


def _bc_bnl(M, name):
    if M.dim() == 4:
        if M.shape[1] != 1:
            raise ValueError(f"{name}: only ngroups == 1 is supported (got {tuple(M.shape)})")
        return M[:, 0], True
    if M.dim() != 3:
        raise ValueError(f"{name} must be (batch, dstate, seqlen) or (batch, 1, dstate, seqlen)")
    return M, False


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None,
                delta_softplus=False, return_last_state=False, grad_mode=True):
        _lib.require_gpu(u, "selective_scan_fn")
        lib = _lib.load()
        io = u.dtype
        code = _lib.dtype_code(io)
        if A.is_complex():
            raise NotImplementedError("complex A is not on the SI-Mamba path")
        B3, ctx.b_grouped = _bc_bnl(B, "B")
        C3, ctx.c_grouped = _bc_bnl(C, "C")
        batch, dim, L = u.shape
        N = A.shape[1]
        if A.shape[0] != dim or B3.shape != (batch, N, L) or C3.shape != (batch, N, L):
            raise ValueError("selective_scan_fn: inconsistent shapes")
        uc = u.contiguous()
        dc = delta.to(io).contiguous()
        Ac = A.float().contiguous()
        # B, C are read through their strides (no transposition copy) when both share them
        Bc, Cc = B3.to(io), C3.to(io)
        if Bc.stride() != Cc.stride() or min(Bc.stride()) < 0:
            Bc, Cc = Bc.contiguous(), Cc.contiguous()
        Dc = None if D is None else D.float().contiguous()
        zc = None if z is None else z.to(io)
        if zc is not None and not _rows_contiguous(zc):
            zc = zc.contiguous()
        bc = None if delta_bias is None else delta_bias.float().contiguous()
        out = torch.empty_like(uc)
        # ctx.needs_input_grad is True under torch.no_grad() too: the caller's grad mode comes in as an argument
        needs_grad = grad_mode and any(ctx.needs_input_grad)
        aligned = all(t is None or t.data_ptr() % 16 == 0 for t in (uc, dc, Bc, Cc, zc))
        ckpt_step, x_ckpt = _lib.scan_plan(batch, dim, L, N, io, aligned and bool(delta_softplus), u.device, needs_grad)
        last = (torch.empty(batch, dim, N, device=u.device, dtype=torch.float32)
                if return_last_state else None)
        with torch.cuda.device(u.device), _lib.timed("scan_fwd", u.device):
            rc = lib.simamba_selective_scan_fwd(
                _lib.ptr(uc), _lib.ptr(dc), _lib.ptr(Ac), _lib.ptr(Bc), _lib.ptr(Cc), _lib.ptr(Dc),
                _lib.ptr(zc), _lib.ptr(bc), _lib.ptr(out), _lib.ptr(x_ckpt), _lib.ptr(last),
                batch, dim, L, N, code, int(bool(delta_softplus)),
                0 if zc is None else zc.stride(0), Bc.stride(0), Bc.stride(1), Bc.stride(2),
                ckpt_step, _lib.current_scan_variant(), _lib.stream_ptr(u.device))
        _lib.check(rc, "simamba_selective_scan_fwd")
        ctx.delta_softplus = bool(delta_softplus)
        ctx.ckpt_step = ckpt_step
        ctx.has = (D is not None, z is not None, delta_bias is not None)
        ctx.in_dtypes = (delta.dtype, B.dtype, C.dtype,
                         None if D is None else D.dtype,
                         None if z is None else z.dtype,
                         None if delta_bias is None else delta_bias.dtype, A.dtype)
        ctx.save_for_backward(uc, dc, Ac, Bc, Cc, Dc, zc, bc, x_ckpt)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        uc, dc, Ac, Bc, Cc, Dc, zc, bc, x_ckpt = ctx.saved_tensors
        lib = _lib.load()
        batch, dim, L = uc.shape
        N = Ac.shape[1]
        io = uc.dtype
        dout = dout.to(io).contiguous()
        du = torch.empty_like(uc)
        ddelta = torch.empty_like(uc)
        dz = torch.empty_like(uc) if zc is not None else None
        dA, dB, dC, dD, dbias = _lib.scan_bwd_accumulators(batch, dim, L, N, Dc is not None, bc is not None,
                                                           uc.device)
        with torch.cuda.device(uc.device), _lib.timed("scan_bwd", uc.device):
            rc = lib.simamba_selective_scan_bwd(
                _lib.ptr(uc), _lib.ptr(dc), _lib.ptr(Ac), _lib.ptr(Bc), _lib.ptr(Cc), _lib.ptr(Dc),
                _lib.ptr(zc), _lib.ptr(bc), _lib.ptr(dout), _lib.ptr(x_ckpt),
                _lib.ptr(du), _lib.ptr(ddelta), _lib.ptr(dA), _lib.ptr(dB), _lib.ptr(dC), _lib.ptr(dD),
                _lib.ptr(dz), _lib.ptr(dbias), batch, dim, L, N, _lib.dtype_code(io),
                int(ctx.delta_softplus), 0 if zc is None else zc.stride(0), 0,
                Bc.stride(0), Bc.stride(1), Bc.stride(2), ctx.ckpt_step, _lib.stream_ptr(uc.device))
        _lib.check(rc, "simamba_selective_scan_bwd")
        dt_delta, dt_B, dt_C, dt_D, dt_z, dt_bias, dt_A = ctx.in_dtypes
        dB = dB.to(dt_B)
        dC = dC.to(dt_C)
        if ctx.b_grouped:
            dB = dB.unsqueeze(1)
        if ctx.c_grouped:
            dC = dC.unsqueeze(1)
        return (du, ddelta.to(dt_delta), dA.to(dt_A), dB, dC,
                None if dD is None else dD.to(dt_D),
                None if dz is None else dz.to(dt_z),
                None if dbias is None else dbias.to(dt_bias),
                None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                      delta_softplus=False, return_last_state=False):
    """u, delta, z: (B, D, L); A: (D, N); B, C: (B, N, L) or (B, 1, N, L); D, delta_bias: (D).

    Returns out (B, D, L) in u's dtype, plus last_state (B, D, N) fp32 when asked.
    """
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state,
                                 torch.is_grad_enabled())
