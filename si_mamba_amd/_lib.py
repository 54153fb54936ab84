"""ctypes binding of libsimamba_hip.so (C ABI: include/simamba.h).

The product path has NO fallback: if the shared library is missing or a symbol
is absent, importing an op raises.  Nothing here (or anywhere in this package)
imports ``oracle``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsimamba_hip.so")

F32, BF16 = 0, 1

SPEC_SYMMETRIC = 0x01
SPEC_SELF_LOOP = 0x02
SPEC_BINARY = 0x04
SPEC_MATRIX_SYM = 0x08
SPEC_SMALLEST = 0x10
SPEC_SIGMA_MEAN = 0x20

# forward-scan kernel selection (include/simamba.h): AUTO in production, the others for benchmarks / parity tests
SCAN_AUTO, SCAN_ROWSCAN, SCAN_LPC2, SCAN_LPC4, SCAN_MIX = 0, 1, 2, 4, 6
# checkpoint layouts handed from the forward scan to the backward (= which backward kernel runs)
CKPT_ROW, CKPT_SEQ = 128, 16

# name -> (restype, argtypes); mirrors include/simamba.h one to one
_P = c_void_p
_LL = c_longlong
ABI_VERSION = 9
SIGNATURES = {
    "simamba_abi_version": (c_int, []),
    "simamba_strerror": (c_char_p, [c_int]),
    "simamba_scan_num_chunks": (c_int, [c_int]),
    "simamba_scan_fwd_auto_variant": (c_int, [c_int, c_int]),
    "simamba_scan_ckpt_step": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "simamba_scan_ckpt_floats": (_LL, [c_int, c_int, c_int, c_int, c_int]),
    "simamba_selective_scan_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                           c_int, c_int, c_int, c_int, c_int, c_int,
                                           _LL, _LL, _LL, _LL, c_int, c_int, _P]),
    "simamba_selective_scan_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                           _P, _P, _P, _P, _P, _P, _P, _P,
                                           c_int, c_int, c_int, c_int, c_int, c_int,
                                           _LL, _LL, _LL, _LL, _LL, c_int, _P]),
    "simamba_selective_scan_dt_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                              c_int, c_int, c_int, c_int, c_int, c_int,
                                              _LL, _LL, _LL, c_int, c_int, _P]),
    "simamba_selective_scan_dt_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P,
                                              _P, _P, _P, _P, _P, _P, _P, _P,
                                              c_int, c_int, c_int, c_int, c_int, c_int,
                                              _LL, _LL, _LL, _LL, _P]),
    "simamba_xdt_proj_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _LL, _P]),
    "simamba_conv_xdt_proj_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _LL,
                                          _P]),
    "simamba_seq_gather_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _LL, c_int, _P]),
    "simamba_seq_gather_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _LL, c_int, _P]),
    "simamba_causal_conv1d_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _LL, _P]),
    "simamba_causal_conv1d_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P,
                                          c_int, c_int, c_int, c_int, c_int, c_int, _LL, _LL, _P]),
    "simamba_add_layer_norm_grid": (c_int, [c_int, c_int]),
    "simamba_add_layer_norm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float,
                                           c_int, c_int, _P]),
    "simamba_add_layer_norm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int,
                                           c_int, c_int, _P]),
    "simamba_out_proj_add_ln_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float,
                                            c_int, _P]),
    "simamba_in_proj_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "simamba_bn_relu_grid": (c_int, [_LL]),
    "simamba_bn_relu_fwd": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _LL, c_int,
                                    _LL, c_int, _P]),
    "simamba_bn_relu_bwd": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _LL, c_int, _LL,
                                    c_int, c_int, _P]),
    "simamba_group_max_fwd": (c_int, [_P, _P, _P, _LL, c_int, c_int, c_int, _P]),
    "simamba_group_max_bwd": (c_int, [_P, _P, _P, _LL, c_int, c_int, c_int, _P]),
    "simamba_three_nn": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "simamba_three_interpolate_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "simamba_three_interpolate_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "simamba_chamfer_fwd": (c_int, [_P, _P, _P, _P, _P, _LL, c_int, c_int, _P]),
    "simamba_chamfer_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _LL, c_int, c_int, _P]),
    "simamba_knn_group": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "simamba_knn_graph": (c_int, [_P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_float, c_uint, _P]),
    "simamba_laplacian_topk": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_uint, _P]),
    "simamba_spectral_workspace_bytes": (c_size_t, [c_int, c_int]),
    "simamba_spectral_topk": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_float,
                                      c_int, c_uint, _P]),
    "simamba_argsort_rows": (c_int, [_P, _P, c_int, c_int, _P]),
    "simamba_farthest_point_sample": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
}

_lib = None


def load():
    """Load the library once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C si_mamba_amd/csrc`.  si_mamba_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table disagree
        fn.restype = res
        fn.argtypes = args
    got = lib.simamba_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libsimamba_hip.so ABI version {got}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().simamba_strerror(rc).decode()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def dtype_code(dtype):
    import torch
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"si_mamba_amd kernels take float32 or bfloat16 activations, got {dtype}")


def stream_ptr(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on {t.device}; si_mamba_amd runs on a ROCm device only "
            "(no CPU fallback -- the CPU restatement lives in oracle/ and is test-only)")


# ---- optional per-kernel device timing (used by bench.py only) -----------------------------------
# When enabled, each C-ABI launch is bracketed by two events recorded on the stream it is enqueued
# on; nothing synchronises until `kernel_times()` is read after the timed region.
_timing = {"on": False, "events": {}, "only": None}


def enable_kernel_timing(on=True, only=None):
    """``only``: an iterable of kernel names to time (the others run without events).  A pair of event records per
    launch is not free: timing all ~100 launches per block stack costs the fp32 step 0.5 ms and makes the bf16 step
    CPU-bound (34 -> 41 ms), so bench.py times only the scan kernels inside its timed region."""
    _timing["on"] = bool(on)
    _timing["events"] = {}
    _timing["only"] = None if only is None else frozenset(only)


class timed:
    __slots__ = ("name", "dev", "ev")

    def __init__(self, name, device):
        self.name, self.dev, self.ev = name, device, None

    def __enter__(self):
        if _timing["on"] and (_timing["only"] is None or self.name in _timing["only"]):
            import torch
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream(self.dev))
            self.ev = (a, b)
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            import torch
            self.ev[1].record(torch.cuda.current_stream(self.dev))
            _timing["events"].setdefault(self.name, []).append(self.ev)
        return False


def kernel_times():
    """name -> (launches, mean ms); call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in _timing["events"].items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    return out


_scan_variant = [SCAN_AUTO]


class scan_variant:
    """Context manager for benchmarks and parity tests: the forward-scan kernel the Python ops request from the
    library inside the block (an explicit argument of simamba_selective_scan_fwd -- the library itself holds no
    state and reads no environment).  Production code never enters it: SCAN_AUTO lets the library choose."""

    def __init__(self, variant):
        self.v, self.prev = int(variant), None

    def __enter__(self):
        self.prev, _scan_variant[0] = _scan_variant[0], self.v
        return self

    def __exit__(self, *exc):
        _scan_variant[0] = self.prev
        return False


def current_scan_variant():
    return _scan_variant[0]


_scan_ckpt = [0]


class scan_ckpt:
    """Context manager for benchmarks and parity tests: the checkpoint layout (CKPT_ROW / CKPT_SEQ, i.e. the backward
    kernel) the Python ops request instead of the library's own choice.  Production code never enters it."""

    def __init__(self, step):
        self.v, self.prev = int(step), None

    def __enter__(self):
        self.prev, _scan_ckpt[0] = _scan_ckpt[0], self.v
        return self

    def __exit__(self, *exc):
        _scan_ckpt[0] = self.prev
        return False


_fuse_dt = [None]      # None: the measured default (bf16 I/O only, see fuse_dt_enabled); True / False: forced
counters = {}          # name -> how often a route was taken (tests assert on them; nothing in the product reads them)


def count(name):
    counters[name] = counters.get(name, 0) + 1



class scan_fuse_dt:
    """Context manager for benchmarks and parity tests: force the mixer to let the scan kernels form delta themselves
    (simamba_selective_scan_dt_fwd / _bwd) wherever they can (True), or to materialise it as upstream does (False).
    Production never enters it and gets the measured default of fuse_dt_enabled()."""

    def __init__(self, on):
        self.v, self.prev = (None if on is None else bool(on)), None

    def __enter__(self):
        self.prev, _fuse_dt[0] = _fuse_dt[0], self.v
        return self

    def __exit__(self, *exc):
        _fuse_dt[0] = self.prev
        return False


def fuse_dt_enabled(dtype):
    """Default: bf16 I/O only.  Measured at (64, 768, 1024), per layer, conv + x_proj (+ dt_proj) / scan forward / scan
    backward (tools/bench_dt_fusion.py, profiles/r03b_dt_fusion.txt): bf16 96 + 315 + 691 us with delta materialised,
    62 + 314 + 691 us formed in the scans (two bf16 MFMAs per tile: free); fp32 168 + 308 + 691 us against 126 + 356 +
    790 us -- the exact-fp32 MFMA runs at the vector rate, and inside the VALU-bound scans its 12 instructions per tile
    are additive, while inside the xdt kernel (bound by bytes in flight) they are hidden."""
    import torch
    return (dtype == torch.bfloat16) if _fuse_dt[0] is None else _fuse_dt[0]


_fuse_out_norm = [True]
_hand_in_proj = [None]   # None: the measured default (in_proj_hand_enabled); True / False: forced


class fuse_out_norm:
    """Context manager for benchmarks and parity tests: let MixerModel.forward apply out_proj fused with the next block's
    add + LayerNorm (out_norm.py; the default wherever the kernel's shapes apply) or op by op as the reference does."""

    def __init__(self, on):
        self.v, self.prev = bool(on), None

    def __enter__(self):
        self.prev, _fuse_out_norm[0] = _fuse_out_norm[0], self.v
        return self

    def __exit__(self, *exc):
        _fuse_out_norm[0] = self.prev
        return False


def fuse_out_norm_enabled():
    return _fuse_out_norm[0]


class hand_in_proj:
    """Context manager for benchmarks and parity tests: in_proj through the hand-written bf16 kernel
    (simamba_in_proj_fwd) wherever its shapes apply (True) or through the library GEMM (False)."""

    def __init__(self, on):
        self.v, self.prev = (None if on is None else bool(on)), None

    def __enter__(self):
        self.prev, _hand_in_proj[0] = _hand_in_proj[0], self.v
        return self

    def __exit__(self, *exc):
        _hand_in_proj[0] = self.prev
        return False


def in_proj_hand_enabled(workgroups, dtype=None):
    """Default: bf16 only (102 us against 116 us for the library GEMM; the fp32 form measured 650 against 535 us and stays an
    explicit variant), and only grids that fill the chip (a workgroup owns 256 tokens of one sample, alone on its CU)."""
    import torch
    if _hand_in_proj[0] is not None:
        return _hand_in_proj[0]
    return dtype != torch.float32 and workgroups >= IN_PROJ_MIN_WORKGROUPS


IN_PROJ_MIN_WORKGROUPS = 192


def scan_plan(batch, dim, seqlen, dstate, dtype, aligned, device, need_grad):
    """(ckpt_step, x_ckpt or None) for one forward / backward pair.  ``aligned``: the caller's statement that every
    activation operand is 16-byte aligned with pack-aligned strides (what the sequential backward needs; the library
    checks again and refuses otherwise)."""
    import torch
    lib = load()
    code = dtype_code(dtype)
    step = _scan_ckpt[0] or lib.simamba_scan_ckpt_step(batch, dim, seqlen, dstate, code)
    if step == CKPT_SEQ and not (aligned and dstate == 16 and dim % 64 == 0 and
                                 seqlen % (4 if code == F32 else 8) == 0):
        step = CKPT_ROW
    if _scan_variant[0] == SCAN_ROWSCAN:
        step = CKPT_ROW                 # an explicit row-scan forward writes 128-step checkpoints only
    if not need_grad:
        return step, None
    n = lib.simamba_scan_ckpt_floats(batch, dim, seqlen, dstate, step)
    return step, (torch.empty(n, device=device, dtype=torch.float32) if n else None)


def scan_bwd_accumulators(batch, dim, seqlen, dstate, has_D, has_bias, device):
    """(dA (D,N), dB (B,N,L), dC (B,N,L), dD (D)|None, ddelta_bias (D)|None): the fp32 accumulators
    simamba_selective_scan_bwd adds into, carved back to back (no padding between them) out of ONE allocation.
    The library zeroes exactly-adjacent spans with one memset node; it never writes outside the spans it is
    given, so separately allocated buffers are equally valid, only slower to clear (include/simamba.h)."""
    import torch
    sizes = [dim * dstate, batch * dstate * seqlen, batch * dstate * seqlen, dim if has_D else 0,
             dim if has_bias else 0]
    flat = torch.empty(sum(sizes), device=device, dtype=torch.float32)
    parts, o = [], 0
    for n in sizes:
        parts.append(flat[o:o + n])
        o += n
    dA = parts[0].view(dim, dstate)
    dB = parts[1].view(batch, dstate, seqlen)
    dC = parts[2].view(batch, dstate, seqlen)
    return dA, dB, dC, (parts[3] if has_D else None), (parts[4] if has_bias else None)
