"""CPU: the C-ABI library loads, exports every symbol include/simamba.h declares, and validates
arguments before touching the device (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from si_mamba_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "simamba.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(simamba_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_version_strerror_and_chunks():
    lib = _lib.load()
    assert lib.simamba_abi_version() == 9
    assert lib.simamba_strerror(0) == b"ok"
    assert b"dstate" in lib.simamba_strerror(-4)
    assert lib.simamba_scan_num_chunks(64) == 1
    assert lib.simamba_scan_num_chunks(128) == 1
    assert lib.simamba_scan_num_chunks(129) == 2
    assert lib.simamba_scan_num_chunks(1024) == 8
    # host-side kernel choice: row-scan below 49 152 rows, two lanes per channel from there on
    assert lib.simamba_scan_fwd_auto_variant(32, 768) == 1
    assert lib.simamba_scan_fwd_auto_variant(64, 768) == 2
    assert lib.simamba_scan_fwd_auto_variant(256, 768) == 2
    # checkpoint plan: the sequential backward (16-step checkpoints) from the row count on at which the forward is a
    # lanes-per-channel kernel, for shapes it can take; sizes of both layouts
    assert lib.simamba_scan_ckpt_step(64, 768, 1024, 16, 0) == 16
    assert lib.simamba_scan_ckpt_step(32, 768, 1024, 16, 0) == 128
    assert lib.simamba_scan_ckpt_step(64, 768, 1024, 8, 0) == 128
    assert lib.simamba_scan_ckpt_step(64, 776, 1024, 16, 0) == 128      # dim % 64
    assert lib.simamba_scan_ckpt_step(64, 768, 1022, 16, 0) == 128      # rows not 16-byte aligned
    assert lib.simamba_scan_ckpt_step(64, 768, 1020, 16, 1) == 128      # bf16: 8-element packs
    assert lib.simamba_scan_ckpt_floats(64, 768, 1024, 16, 16) == 64 * 64 * 768 * 16
    assert lib.simamba_scan_ckpt_floats(64, 768, 1024, 16, 128) == 64 * 768 * 8 * 16
    assert lib.simamba_scan_ckpt_floats(64, 768, 16, 16, 16) == 0
    assert lib.simamba_scan_ckpt_floats(64, 768, 128, 16, 128) == 0
    assert lib.simamba_spectral_workspace_bytes(4, 128) == 256 + 4 * 128 * 128 * 4
    assert b"variant" in lib.simamba_strerror(-9)


def test_argument_validation_precedes_any_launch():
    lib = _lib.load()
    n = None
    one = ctypes.c_void_p(16)   # never dereferenced: every call below fails validation first
    assert lib.simamba_selective_scan_fwd(n, n, n, n, n, n, n, n, n, n, n, 1, 1, 1, 16, 0, 1, 0, 0, 0, 0, 0, 0, n) == -1
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, n, n, 1, 8, 8, 17, 0, 1, 0, 0, 0, 0, 0, 0, n) == -4
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, n, n, 1, 8, 8, 16, 7, 1, 0, 0, 0, 0, 0, 0, n) == -3
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, n, n, 0, 8, 8, 16, 0, 1, 0, 0, 0, 0, 0, 0, n) == 0
    # an unknown kernel variant, and an explicit lanes-per-channel request the shape cannot take (dstate != 16)
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, n, n, 1, 8, 8, 16, 0, 1, 0, 0, 0, 0, 0, 3, n) == -9
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, n, n, 1, 8, 8, 8, 0, 1, 0, 0, 0, 0, 0, 2, n) == -9
    # a checkpoint layout that does not exist; 16-step checkpoints from the row-scan kernel
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, one, n, 1, 64, 64, 16, 0, 1, 0, 0, 0, 0, 32, 0, n) == -9
    assert lib.simamba_selective_scan_fwd(one, one, one, one, one, n, n, n, one, one, n, 1, 64, 64, 16, 0, 1, 0, 0, 0, 0, 16, 1, n) == -9
    assert lib.simamba_causal_conv1d_fwd(one, one, n, one, 1, 8, 8, 5, 1, 0, 0, n) == -5
    assert lib.simamba_laplacian_topk(one, n, n, n, n, n, 1, 129, 4, 0, n) == -7
    assert lib.simamba_knn_graph(one, one, n, 0, 1, 16, 3, 16, 1.0, 0, n) == -7
    assert lib.simamba_spectral_topk(one, n, n, n, one, 8, 1, 16, 4, 1.0, 4, 0, n) == -6
    assert lib.simamba_argsort_rows(one, one, 1, 2048, n) == -2


def test_product_path_refuses_cpu_tensors():
    import torch
    from si_mamba_amd import selective_scan_fn, causal_conv1d_fn, Mamba
    u = torch.zeros(1, 4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        selective_scan_fn(u, u, torch.zeros(4, 2), torch.zeros(1, 2, 8), torch.zeros(1, 2, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        causal_conv1d_fn(u, torch.zeros(4, 4))
    with pytest.raises(RuntimeError):
        Mamba(16)(torch.zeros(1, 8, 16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "si_mamba_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
