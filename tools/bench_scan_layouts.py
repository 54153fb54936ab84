"""Forward-scan variants on the mixer's operand layouts (z = half of the in_proj output, B | C token-major inside
x_dbl) against contiguous operands; trains of 20 back-to-back calls (tuning tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib, selective_scan_fn
from si_mamba_amd.synthetic import scan_inputs

dev = torch.device("cuda:0")
B, D, L, N, R = 64, 768, 1024, 16, 24
t = {k: (v.to(dev) if v is not None else None) for k, v in scan_inputs(B, D, L, N, seed=0).items()}
xz = torch.randn(B, 2 * D, L, device=dev)
xz[:, D:] = t["z"]
x_dbl = torch.randn(B, L, R + 2 * N, device=dev)
x_dbl[:, :, R:R + N] = t["B"].transpose(1, 2)
x_dbl[:, :, R + N:] = t["C"].transpose(1, 2)
layouts = {
    "contiguous": (t["z"], t["B"], t["C"]),
    "z strided": (xz[:, D:], t["B"], t["C"]),
    "B|C token-major": (t["z"], x_dbl[:, :, R:R + N].transpose(1, 2), x_dbl[:, :, R + N:].transpose(1, 2)),
    "mixer (both)": (xz[:, D:], x_dbl[:, :, R:R + N].transpose(1, 2), x_dbl[:, :, R + N:].transpose(1, 2)),
}
for name, (z, Bm, Cm) in layouts.items():
    for variant in (2, 6, 4, 2, 6):
        _lib._scan_variant[0] = variant
        with torch.no_grad():
            for _ in range(3):
                selective_scan_fn(t["u"], t["delta"], t["A"], Bm, Cm, t["D"], z, t["delta_bias"], True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                selective_scan_fn(t["u"], t["delta"], t["A"], Bm, Cm, t["D"], z, t["delta_bias"], True)
            b.record()
            torch.cuda.synchronize()
        print(f"{name:18s} variant {variant}: {a.elapsed_time(b) * 1e3 / 20:7.1f} us", flush=True)
