"""si_mamba_amd -- MI355X-native SI-Mamba hot path (selective scan, causal conv1d, spectral ordering).

Importing this package does not load the HIP library; the first op call does, and raises if
``libsimamba_hip.so`` is missing (there is no CPU fallback -- see DESIGN.md).
"""
from .causal_conv1d import causal_conv1d_fn
from .selective_scan import selective_scan_fn
from .mamba_simple import Mamba
from .block import Block, MixerModel, create_block, DropPath
from . import spectral
from .shim import install_shim

__all__ = ["causal_conv1d_fn", "selective_scan_fn", "Mamba", "Block", "MixerModel", "create_block",
           "DropPath", "spectral", "install_shim"]
