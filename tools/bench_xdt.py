"""x_proj -> dt_proj: the fused MFMA kernel against the two library GEMMs it replaces (tuning tool, fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd.gemm_tuning import enable_tuned_gemms
from si_mamba_amd.mamba_inner import _wx, _xw, xdt_proj_fwd

dev = torch.device("cuda:0")
enable_tuned_gemms()
for B, D, L, N, R in [(64, 768, 1024, 16, 24), (128, 768, 1024, 16, 24), (64, 768, 512, 16, 24)]:
    S = R + 2 * N
    x = torch.randn(B, D, L, device=dev)
    wx = torch.randn(S, D, device=dev) / D ** 0.5
    wdt = torch.randn(D, R, device=dev) / R ** 0.5

    def lib():
        x_dbl = _xw(x.transpose(1, 2), wx.t())
        return x_dbl, _wx(wdt, x_dbl[:, :, :R].transpose(1, 2))

    def fused():
        return xdt_proj_fwd(x, wx, wdt)

    for name, fn in (("library (2 GEMMs)", lib), ("fused MFMA kernel", fused)):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        flops = 2.0 * B * L * (S * D + D * R)
        nbytes = 4.0 * B * L * (2 * D + S)
        print(f"({B},{D},{L}) {name:20s} {us:8.1f} us  {flops / us * 1e-6:6.1f} TF/s  {nbytes / us * 1e-3:7.1f} GB/s", flush=True)
    a, b = lib(), fused()
    print("   max |diff| x_dbl %.2e delta %.2e" % ((a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item()))

# the conv-fused form against conv kernel + fused projections
from si_mamba_amd import causal_conv1d_fn
for B, D, L in [(64, 768, 1024), (128, 768, 1024)]:
    xz = torch.randn(B, 2 * D, L, device=dev)
    x_in = xz[:, :D]
    cw = torch.randn(D, 4, device=dev) * 0.5
    cb = torch.randn(D, device=dev)
    wx = torch.randn(56, D, device=dev) / D ** 0.5
    wdt = torch.randn(D, 24, device=dev) / 24 ** 0.5
    xc = torch.empty(B, D, L, device=dev)
    for name, fn in (("conv kernel + xdt kernel", lambda: xdt_proj_fwd(causal_conv1d_fn(x_in, cw, cb, "silu"), wx, wdt)),
                     ("conv fused into xdt", lambda: xdt_proj_fwd(x_in, wx, wdt, conv=(cw, cb, xc)))):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"({B},{D},{L}) {name:26s} {a.elapsed_time(b) * 1e3 / 20:8.1f} us", flush=True)
