"""Drop-in for ``mamba_ssm.modules.mamba_simple.Mamba`` on MI355X.

The reference builds its mixer as ``partial(Mamba, layer_idx=i, **ssm_cfg, device=None,
dtype=None)`` (models/point_mamba.py:162), instantiates it as ``mixer_cls(dim)``
(models/block.py:36) and calls ``self.mixer(hidden_states, inference_params=...)``
(models/block.py:72).  This class keeps that constructor signature, forward signature and
the exact parameter names/shapes of the reference's checkpoints
(logs/finetuned_hardest.log:132-148):

    A_log (D,N)  D (D)  in_proj.weight (2D,d)  conv1d.weight (D,1,W)  conv1d.bias (D)
    x_proj.weight (R+2N,D)  dt_proj.weight (D,R)  dt_proj.bias (D)  out_proj.weight (d,D)

Data flow: in_proj -> causal conv1d + SiLU (HIP) -> x_proj ->
dt_proj -> selective scan with fused softplus/D-skip/SiLU gate (HIP) -> out_proj.  The four
projections are plain library GEMMs through PyTorch-ROCm.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .causal_conv1d import causal_conv1d_fn
from .mamba_inner import in_proj_fn, mamba_inner_fn
from .selective_scan import selective_scan_fn


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001,
                 dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True,
                 bias=False, use_fast_path=True, layer_idx=None, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        if not 1 <= d_state <= 16:
            raise ValueError("si_mamba_amd.Mamba: d_state must be in [1,16] (the reference uses 16)")
        if not 2 <= d_conv <= 4:
            raise ValueError("si_mamba_amd.Mamba: d_conv must be in [2,4] (the reference uses 4)")

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.conv1d = nn.Conv1d(in_channels=self.d_inner, out_channels=self.d_inner, bias=conv_bias,
                                kernel_size=d_conv, groups=self.d_inner, padding=d_conv - 1,
                                **factory_kwargs)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)

        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        # dt bias such that softplus(bias) is log-uniform in [dt_min, dt_max]
        dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        # the reference's _init_weights (models/point_mamba.py:122-125) must not zero this bias
        self.dt_proj.bias._no_reinit = True

        A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)

    def forward(self, hidden_states, inference_params=None, A=None):
        """hidden_states: (B, L, d_model) -> (B, L, d_model).  ``A`` (optional, beyond upstream's signature):
        -exp(A_log) when the caller has already formed it (block.MixerModel does, for all layers at once)."""
        if inference_params is not None:
            raise NotImplementedError(
                "step-wise decoding caches are not on the SI-Mamba path (no reference runner passes "
                "inference_params; models/block.py:75-76 is never reached)")
        batch, seqlen, _ = hidden_states.shape
        # (2D, d) @ (B, d, L) -> (B, 2D, L): L is the contiguous axis the HIP kernels stream along
        if self.use_fast_path:
            return self.forward_xz(self.in_proj_xz(hidden_states), A=A)
        if A is None:
            A = -torch.exp(self.A_log.float())
        # reference composition of the separate ops (same kernels; torch.matmul / chunk copy the views)
        xz = torch.matmul(self.in_proj.weight, hidden_states.transpose(1, 2))
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[None, :, None]
        x, z = xz.chunk(2, dim=1)
        x = causal_conv1d_fn(x, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_dbl = torch.matmul(x.transpose(1, 2), self.x_proj.weight.t())            # (B, L, R+2N)
        dt, Bm, Cm = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        delta = torch.matmul(self.dt_proj.weight, dt.transpose(1, 2))              # (B, D, L), bias in scan
        y = selective_scan_fn(x, delta, A, Bm.transpose(1, 2), Cm.transpose(1, 2), self.D.float(), z=z,
                              delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
        out = torch.matmul(y.transpose(1, 2), self.out_proj.weight.t())
        if self.out_proj.bias is not None:
            out = out + self.out_proj.bias.to(out.dtype)
        return out

    def in_proj_xz(self, hidden_states):
        """(B, L, d_model) -> xz (B, 2 d_inner, L), L contiguous: the in_proj half of forward()."""
        return in_proj_fn(hidden_states, self.in_proj.weight, self.in_proj.bias)

    def forward_xz(self, xz, A=None, emit_y=False):
        """xz (B, 2 d_inner, L) -> (B, L, d_model): everything after in_proj as one autograd node, no
        activation-sized copies (mamba_inner.py).  forward(h) == forward_xz(in_proj_xz(h)); MixerModel calls the two
        halves separately for the first block when the sequence is an expansion of fewer distinct tokens
        (seq_expand.py).  ``emit_y=True``: stop before out_proj and return the gated scan output (B, d_inner, L) --
        MixerModel then applies out_proj fused with the next block's add + LayerNorm (out_norm.py)."""
        if A is None:                                       # MixerModel hands over -exp(A_log) of all its layers at once
            A = -torch.exp(self.A_log.float())
        return mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                              self.dt_proj.weight, None if emit_y else self.out_proj.weight,
                              None if emit_y else self.out_proj.bias, A,
                              self.D.float(), delta_bias=self.dt_proj.bias.float(),
                              dt_rank=self.dt_rank, d_state=self.d_state)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError("inference caches are outside the SI-Mamba hot path")
