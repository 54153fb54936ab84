"""CPU restatement of the selective-scan / causal-conv1d / Mamba-mixer arithmetic.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity unpinned by the
reference: the arithmetic is in the absent wheels mamba-ssm / causal-conv1d
(reference README.md:55-56); the reference reaches it only through
``models/block.py:72`` (``self.mixer(hidden_states, ...)``) with the mixer
built at ``models/point_mamba.py:162``.  What is restated here is the published
algorithm of those packages:

    delta = softplus(delta_raw + delta_bias)
    h_t   = exp(delta_t * A) * h_{t-1} + (delta_t * u_t) * B_t
    y_t   = <h_t, C_t> + D * u_t ;   out = y * silu(z)

Everything accumulates in fp32 (or fp64 when asked) regardless of I/O dtype.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _bc_to_bnl(M: torch.Tensor) -> torch.Tensor:
    """Accept (B,N,L) or the grouped (B,1,N,L) form the upstream wrapper uses."""
    if M.dim() == 4:
        if M.shape[1] != 1:
            raise ValueError("only ngroups == 1 is on the SI-Mamba path")
        M = M[:, 0]
    return M


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                       delta_softplus=False, return_last_state=False,
                       acc_dtype=torch.float32):
    """Sequential-over-L scan.  u, delta, z: (B,D,L); A: (D,N); B, C: (B,N,L)."""
    io_dtype = u.dtype
    B = _bc_to_bnl(B).to(acc_dtype)
    C = _bc_to_bnl(C).to(acc_dtype)
    uf = u.to(acc_dtype)
    dt = delta.to(acc_dtype)
    if delta_bias is not None:
        dt = dt + delta_bias.to(acc_dtype)[None, :, None]
    if delta_softplus:
        dt = F.softplus(dt)
    Af = A.to(acc_dtype)
    bsz, dim, L = uf.shape
    n = Af.shape[1]
    h = torch.zeros(bsz, dim, n, dtype=acc_dtype, device=u.device)
    cols = []
    du = dt * uf
    for t in range(L):
        decay = torch.exp(dt[:, :, t, None] * Af[None])            # (B,D,N)
        h = decay * h + du[:, :, t, None] * B[:, None, :, t]
        cols.append((h * C[:, None, :, t]).sum(-1))
    y = torch.stack(cols, dim=2) if cols else uf.new_zeros(bsz, dim, 0)
    if D is not None:
        y = y + uf * D.to(acc_dtype)[None, :, None]
    if z is not None:
        y = y * F.silu(z.to(acc_dtype))
    y = y.to(io_dtype)
    return (y, h) if return_last_state else y


def selective_scan_closed_form(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                               delta_softplus=False):
    """Independent float64 check: h_t = sum_s exp(A * sum_{s<m<=t} delta_m) * delta_s u_s B_s.

    O(L^2) -- small cases only.  Used to cross-check selective_scan_ref.
    """
    f64 = torch.float64
    uf, dt = u.to(f64), delta.to(f64)
    if delta_bias is not None:
        dt = dt + delta_bias.to(f64)[None, :, None]
    if delta_softplus:
        dt = F.softplus(dt)
    Af, Bf, Cf = A.to(f64), _bc_to_bnl(B).to(f64), _bc_to_bnl(C).to(f64)
    cs = torch.cumsum(dt, dim=2)                                     # (B,D,L)
    L = uf.shape[2]
    # gap[b,d,t,s] = sum_{s<m<=t} delta_m  (t >= s)
    gap = cs[:, :, :, None] - cs[:, :, None, :]
    mask = torch.tril(torch.ones(L, L, dtype=torch.bool))
    w = torch.exp(gap[..., None] * Af[None, :, None, None, :])       # (B,D,t,s,N)
    w = w * mask[None, None, :, :, None]
    src = (dt * uf)[:, :, None, :, None] * Bf.permute(0, 2, 1)[:, None, None, :, :]
    h = (w * src).sum(3)                                             # (B,D,t,N)
    y = (h * Cf.permute(0, 2, 1)[:, None]).sum(-1)
    if D is not None:
        y = y + uf * D.to(f64)[None, :, None]
    if z is not None:
        y = y * F.silu(z.to(f64))
    return y


def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """Depthwise causal conv.  x: (B,D,L); weight: (D,W); bias: (D).

    out[b,d,t] = bias[d] + sum_k weight[d,k] * x[b,d,t-(W-1)+k]  (zero left pad),
    optionally followed by SiLU -- the op the mixer applies before x_proj
    (SURVEY.md Appendix A.1).
    """
    if activation not in (None, "silu", "swish"):
        raise ValueError("activation must be None, 'silu' or 'swish'")
    io_dtype = x.dtype
    D, W = weight.shape
    L = x.shape[-1]
    out = F.conv1d(x.float(), weight.float()[:, None, :],
                   None if bias is None else bias.float(), padding=W - 1, groups=D)[..., :L]
    if activation is not None:
        out = F.silu(out)
    return out.to(io_dtype)


class MambaRef(nn.Module):
    """Plain-torch mixer with the parameter set the reference's checkpoints hold.

    Names/shapes: reference logs/finetuned_hardest.log:132-148
    (A_log, D, in_proj.weight, conv1d.weight, conv1d.bias, x_proj.weight,
    dt_proj.weight, dt_proj.bias, out_proj.weight).  Construction arguments are
    those the reference passes at models/point_mamba.py:162 (defaults otherwise).
    """

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto",
                 dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, conv_bias=True, bias=False, layer_idx=None):
        super().__init__()
        self.d_model, self.d_state, self.d_conv = d_model, d_state, d_conv
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.layer_idx = layer_idx
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner,
                                padding=d_conv - 1, bias=conv_bias)
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True)
        std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, std)
        else:
            nn.init.uniform_(self.dt_proj.weight, -std, std)
        dt = torch.exp(torch.rand(self.d_inner) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            self.dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        self.dt_proj.bias._no_reinit = True
        A = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(self.d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias)

    def forward(self, hidden_states, inference_params=None, io_dtype=None):
        """``io_dtype=torch.bfloat16`` restates the mixer as it runs under ``torch.autocast`` (the reference's
        pre-training and segmentation runners, tools/runner_pretrain.py:243): upstream ``mamba_inner_fn`` casts
        the three projection weights (and in_proj goes through autocast's linear) to the autocast dtype, every
        GEMM, the conv and the scan read and write that dtype and accumulate in fp32, while conv1d.weight/bias,
        A, D and dt_proj.bias stay fp32.  Here: values are rounded to ``io_dtype`` at exactly those op
        boundaries and carried as fp32 in between, so a device path with the same roundings differs from this
        only by accumulation order."""
        r = (lambda t: t) if io_dtype is None else (lambda t: t.to(io_dtype).float())
        Bsz, L, _ = hidden_states.shape
        xz = F.linear(r(hidden_states), r(self.in_proj.weight), None if self.in_proj.bias is None
                      else r(self.in_proj.bias))
        xz = r(xz).transpose(1, 2)                                   # (B,2D,L)
        x, z = xz.chunk(2, dim=1)
        x = r(causal_conv1d_ref(x, self.conv1d.weight[:, 0], self.conv1d.bias, "silu"))
        x_dbl = r(F.linear(x.transpose(1, 2), r(self.x_proj.weight)))    # (B,L,R+2N)
        dt, Bm, Cm = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        delta = r(dt @ r(self.dt_proj.weight).t()).transpose(1, 2)  # bias goes into the scan
        y = selective_scan_ref(x, delta, -torch.exp(self.A_log.float()),
                               Bm.transpose(1, 2), Cm.transpose(1, 2), self.D.float(),
                               z=z, delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
        return r(F.linear(r(y).transpose(1, 2), r(self.out_proj.weight),
                          None if self.out_proj.bias is None else r(self.out_proj.bias)))
