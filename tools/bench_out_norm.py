"""out_proj + residual add + LayerNorm at the model shape (B=64, L=1024, d=384, D=768), bf16: the fused kernel
(csrc/out_norm_bf16.hip) against the library GEMM + the add_layer_norm kernel it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
if os.environ.get("SIMAMBA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SIMAMBA_LIB"])
from si_mamba_amd.add_norm import add_layer_norm_fn
from si_mamba_amd.out_norm import out_proj_add_ln_fn

dev = torch.device("cuda:0")
B, C, L = int(os.environ.get("B", 64)), 384, int(os.environ.get("L", 1024))
K = 2 * C
g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn(B, K, L, device=dev, generator=g).bfloat16()
w = torch.randn(C, K, device=dev, generator=g) * K ** -0.5
wb = w.bfloat16()
res = torch.randn(B, L, C, device=dev, generator=g)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)


def fused():
    return out_proj_add_ln_fn(y, w, res, gamma, beta, 1e-5, None, torch.bfloat16)


def unfused():
    hid = torch.bmm(y.transpose(1, 2), wb.t().unsqueeze(0).expand(B, -1, -1))
    return add_layer_norm_fn(hid, res, gamma, beta, 1e-5, out_dtype=torch.bfloat16)


def gemm_only():
    return torch.bmm(y.transpose(1, 2), wb.t().unsqueeze(0).expand(B, -1, -1))


with torch.no_grad():
    for name, fn in (("fused out_proj+add+LN", fused), ("library GEMM + add_ln kernel", unfused), ("library GEMM alone", gemm_only)) * 2:
        for _ in range(3):
            fn()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) / 20 * 1e3
        hbm = (B * K * L * 2 + C * K * 2 + 2 * B * L * C * 4 + B * L * C * 2) / 1e6
        print(f"B={B} L={L} {name:32s} {us:7.1f} us   ({hbm:.0f} MB algorithmic for the fused op -> {hbm / us * 1e-3 * 1e3:.0f} GB/s; "
              f"{2 * B * L * C * K / us * 1e-6:.0f} TF/s)", flush=True)
