"""Seeded synthetic inputs of the shapes the benchmarks and parity tests use.

Plain data generators: no reference to the oracle, usable from bench.py, tools/ and tests alike.
"""
from __future__ import annotations

import math

import torch


def scan_inputs(batch, dim, L, N, seed, with_z=True, with_D=True, with_bias=True):
    """u, z ~ N(0,1); delta_raw ~ N(0, 0.5^2); delta_bias = softplus^-1 of log-uniform[1e-3, 1e-1];
    A = -exp(log(1..N) + N(0, 0.1^2)); B, C ~ N(0,1); D ~ 1 + 0.1 N(0,1); dout ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(batch, dim, L, generator=g)
    z = torch.randn(batch, dim, L, generator=g) if with_z else None
    delta = 0.5 * torch.randn(batch, dim, L, generator=g)
    dt = torch.exp(torch.rand(dim, generator=g) * (math.log(0.1) - math.log(0.001)) + math.log(0.001))
    bias = (dt + torch.log(-torch.expm1(-dt))) if with_bias else None
    A = -torch.exp(torch.log(torch.arange(1, N + 1, dtype=torch.float32))[None, :].repeat(dim, 1)
                   + 0.1 * torch.randn(dim, N, generator=g))
    Bm = torch.randn(batch, N, L, generator=g)
    Cm = torch.randn(batch, N, L, generator=g)
    D = (1.0 + 0.1 * torch.randn(dim, generator=g)) if with_D else None
    dout = torch.randn(batch, dim, L, generator=g)
    return dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, z=z, delta_bias=bias, dout=dout)


def unit_ball_centers(B, G, seed):
    """Gaussian centres, centred and scaled into the unit ball per cloud (pc_norm of the reference's datasets)."""
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, G, 3, generator=g)
    p = p - p.mean(1, keepdim=True)
    return p / p.norm(dim=-1).max(dim=1)[0][:, None, None]


def surface_clouds(B, N, seed):
    """(B, N, 3) clouds sampled on thin closed surfaces (a torus of random aspect per cloud, 2 % radial noise),
    pc_norm'ed: object scans are 2-D sheets, not Gaussian blobs -- their k-NN graphs are far more regular and their
    Laplacians have closer eigenvalue pairs."""
    g = torch.Generator().manual_seed(seed)
    a, b = 2 * torch.pi * torch.rand(B, N, generator=g), 2 * torch.pi * torch.rand(B, N, generator=g)
    r = 0.25 + 0.3 * torch.rand(B, 1, generator=g)
    rad = 1.0 + r * torch.cos(b)
    p = torch.stack([rad * torch.cos(a), rad * torch.sin(a), (0.6 + 0.8 * torch.rand(B, 1, generator=g)) * r * torch.sin(b)], -1)
    p = p * (1.0 + 0.02 * torch.randn(B, N, 1, generator=g))
    p = p - p.mean(1, keepdim=True)
    return p / p.norm(dim=-1).max(dim=1)[0][:, None, None]


def make_clouds(B, N, seed, device="cpu"):
    """(B, N, 3) clouds: N(0, I_3) points, pc_norm (datasets/ShapeNet55Dataset.py:47-53)."""
    return unit_ball_centers(B, N, seed).to(device)
