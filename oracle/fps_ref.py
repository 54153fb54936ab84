"""CPU restatement of farthest-point sampling.  TEST INFRASTRUCTURE (see oracle/__init__.py).

pytorch3d is absent from this container; this restates the published algorithm of
``pytorch3d.ops.sample_farthest_points`` as the reference calls it (models/point_mamba.py:93: lengths=None,
random_start_point=False): start at point 0, repeatedly take the point whose distance to the selected set
is largest (squared Euclidean, first maximum on ties).  Parity unpinned by the reference (no fixtures).
"""
import torch


def sample_farthest_points(points, K):
    B, N, _ = points.shape
    idx = torch.zeros(B, K, dtype=torch.long)
    mind = torch.full((B, N), float("inf"), dtype=points.dtype)
    cur = torch.zeros(B, dtype=torch.long)
    ar = torch.arange(B)
    for i in range(K):
        idx[:, i] = cur
        diff = points - points[ar, cur][:, None, :]
        d = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        mind = torch.minimum(mind, d)
        cur = mind.argmax(dim=1)
    return torch.gather(points, 1, idx.unsqueeze(-1).expand(-1, -1, 3)), idx
