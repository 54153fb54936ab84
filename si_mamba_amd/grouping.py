"""Tokeniser ops ahead of the hot path ("next" row 1).

``sample_farthest_points`` mirrors ``pytorch3d.ops.sample_farthest_points(points, K=...)`` as the reference
calls it (models/point_mamba.py:93): returns ``(centers (B,K,3), idx (B,K))``, first pick = point 0.
"""
from __future__ import annotations

import torch

from . import _lib


def sample_farthest_points(points, K):
    _lib.require_gpu(points, "sample_farthest_points")
    lib = _lib.load()
    p = points.detach().float().contiguous()
    B, N, F = p.shape
    if F != 3:
        raise ValueError("sample_farthest_points expects (B, N, 3)")
    idx = torch.empty(B, K, device=p.device, dtype=torch.int64)
    centers = torch.empty(B, K, 3, device=p.device, dtype=torch.float32)
    with torch.cuda.device(p.device), _lib.timed("fps", p.device):
        rc = lib.simamba_farthest_point_sample(_lib.ptr(p), _lib.ptr(idx), _lib.ptr(centers), B, N, int(K),
                                               _lib.stream_ptr(p.device))
    _lib.check(rc, "simamba_farthest_point_sample")
    return centers.to(points.dtype), idx


def knn_group(centers, points, K):
    """(B,G,3) centres, (B,N,3) points -> (B,G,K) int64: the K nearest points of every centre (the reference's
    pytorch3d.ops.knn_points(center, xyz, K, return_sorted=False).idx at models/point_mamba.py:96)."""
    _lib.require_gpu(points, "knn_group")
    lib = _lib.load()
    p = points.detach().float().contiguous()
    c = centers.detach().float().contiguous()
    B, N, _ = p.shape
    G = c.shape[1]
    idx = torch.empty(B, G, int(K), device=p.device, dtype=torch.int64)
    with torch.cuda.device(p.device), _lib.timed("knn_group", p.device):
        rc = lib.simamba_knn_group(_lib.ptr(p), _lib.ptr(c), _lib.ptr(idx), B, N, G, int(K), _lib.stream_ptr(p.device))
    _lib.check(rc, "simamba_knn_group")
    return idx
