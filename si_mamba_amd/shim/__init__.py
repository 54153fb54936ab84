"""Import shim: makes the reference's import lines resolve to this framework.

The reference does ``from mamba_ssm.modules.mamba_simple import Mamba`` (models/point_mamba.py:25,
part_segmentation/models/pt_mamba.py:16), ``from mamba_ssm.modules.mamba2 import Mamba2``
(models/point_mamba.py:26, imported, never used) and -- inside try/except --
``from mamba_ssm.ops.triton.layernorm import ...`` (models/block.py:9-12).  ``install_shim()``
registers synthetic ``mamba_ssm`` / ``causal_conv1d`` modules in ``sys.modules`` that point at the
HIP-backed implementations; the Triton sub-module is deliberately absent so the reference's
try/except falls back to plain ``nn.LayerNorm`` (the only path its configs use).

``install_shim(pytorch3d=True)`` additionally registers ``pytorch3d.ops`` / ``pytorch3d.loss`` stand-ins for the
three pytorch3d entry points the reference calls (models/point_mamba.py:24, :37;
part_segmentation/models/pt_mamba.py:13): ``sample_farthest_points``, ``knn_points`` and ``chamfer_distance``,
on the HIP kernels, with the call forms and return shapes the reference uses.
"""
from __future__ import annotations

import sys
import types


def install_shim(force: bool = False, pytorch3d: bool = False):
    """Register ``mamba_ssm`` and ``causal_conv1d`` (and, on request, ``pytorch3d``) stand-ins.  Refuses to shadow
    real packages."""
    from .. import causal_conv1d as _cc
    from .. import mamba_simple as _ms
    from .. import selective_scan as _ss

    if not force:
        for name in ("mamba_ssm", "causal_conv1d") + (("pytorch3d",) if pytorch3d else ()):
            mod = sys.modules.get(name)
            if mod is not None and not getattr(mod, "__simamba_shim__", False):
                raise RuntimeError(f"a real '{name}' package is already imported; pass force=True to shadow it")

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__simamba_shim__ = True
        m.__path__ = []          # behaves as a package: sub-imports consult sys.modules first
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class Mamba2:  # imported by the reference, never constructed
        def __init__(self, *a, **kw):
            raise NotImplementedError("Mamba2 is imported but never used by SI-Mamba; not built")

    root = _mod("mamba_ssm", Mamba=_ms.Mamba)
    modules = _mod("mamba_ssm.modules")
    simple = _mod("mamba_ssm.modules.mamba_simple", Mamba=_ms.Mamba)
    m2 = _mod("mamba_ssm.modules.mamba2", Mamba2=Mamba2)
    ops = _mod("mamba_ssm.ops")
    ssi = _mod("mamba_ssm.ops.selective_scan_interface", selective_scan_fn=_ss.selective_scan_fn,
               SelectiveScanFn=_ss.SelectiveScanFn)
    root.modules, root.ops = modules, ops
    modules.mamba_simple, modules.mamba2 = simple, m2
    ops.selective_scan_interface = ssi
    _mod("causal_conv1d", causal_conv1d_fn=_cc.causal_conv1d_fn)
    if pytorch3d:
        _install_pytorch3d(_mod)
    return root


def _install_pytorch3d(_mod):
    """The three pytorch3d calls of the reference, with its argument forms:
         sample_farthest_points(points=xyz, K=G)[0]                      (models/point_mamba.py:93)
         knn_points(center, xyz, K=M, return_sorted=False).idx           (:96)
         chamfer_distance(pred, gt, batch_reduction=None)[0]             (:3203)."""
    import collections

    import torch

    from .. import grouping, mae

    KNN = collections.namedtuple("_KNN", ["dists", "idx", "knn"])

    def sample_farthest_points(points, lengths=None, K=50, random_start_point=False):
        if lengths is not None or random_start_point:
            raise NotImplementedError("sample_farthest_points shim: fixed-length clouds, start at point 0")
        return grouping.sample_farthest_points(points, K)

    def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False,
                   return_sorted=True):
        if lengths1 is not None or lengths2 is not None or norm != 2 or return_nn:
            raise NotImplementedError("knn_points shim: fixed-length clouds, squared L2, indices and distances only")
        idx = grouping.knn_group(p1, p2, K)                                        # ascending: also valid for sorted
        nb = torch.gather(p2.unsqueeze(1).expand(-1, p1.shape[1], -1, -1), 2,
                          idx.unsqueeze(-1).expand(-1, -1, -1, p2.shape[-1]))
        return KNN(dists=((nb - p1.unsqueeze(2)) ** 2).sum(-1), idx=idx, knn=None)

    def chamfer_distance(x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None, weights=None,
                         batch_reduction="mean", point_reduction="mean", norm=2, single_directional=False,
                         abs_cosine=True):
        if any(a is not None for a in (x_lengths, y_lengths, x_normals, y_normals, weights)) or norm != 2 \
                or point_reduction != "mean" or single_directional:
            raise NotImplementedError("chamfer_distance shim: equal-length clouds, squared L2, mean over points")
        d = mae.chamfer_distance(x, y)
        if batch_reduction == "mean":
            d = d.mean()
        elif batch_reduction == "sum":
            d = d.sum()
        return d, None

    root3d = _mod("pytorch3d")
    root3d.ops = _mod("pytorch3d.ops", sample_farthest_points=sample_farthest_points, knn_points=knn_points)
    root3d.loss = _mod("pytorch3d.loss", chamfer_distance=chamfer_distance)
