"""Import shim: makes the reference's import lines resolve to this framework.

The reference does ``from mamba_ssm.modules.mamba_simple import Mamba`` (models/point_mamba.py:25,
part_segmentation/models/pt_mamba.py:16), ``from mamba_ssm.modules.mamba2 import Mamba2``
(models/point_mamba.py:26, imported, never used) and -- inside try/except --
``from mamba_ssm.ops.triton.layernorm import ...`` (models/block.py:9-12).  ``install_shim()``
registers synthetic ``mamba_ssm`` / ``causal_conv1d`` modules in ``sys.modules`` that point at the
HIP-backed implementations; the Triton sub-module is deliberately absent so the reference's
try/except falls back to plain ``nn.LayerNorm`` (the only path its configs use).
"""
from __future__ import annotations

import sys
import types


def install_shim(force: bool = False):
    """Register ``mamba_ssm`` and ``causal_conv1d`` stand-ins.  Refuses to shadow real packages."""
    from .. import causal_conv1d as _cc
    from .. import mamba_simple as _ms
    from .. import selective_scan as _ss

    if not force:
        for name in ("mamba_ssm", "causal_conv1d"):
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
    return root
