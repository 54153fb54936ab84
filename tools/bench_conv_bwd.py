"""Causal conv1d + SiLU backward (csrc/conv1d.hip) at the model shape, fp32 and bf16: time and bytes/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
if os.environ.get("SIMAMBA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SIMAMBA_LIB"])
dev = torch.device("cuda:0")
lib = _lib.load()
B, D, L, W = 64, 768, 1024, 4
for dt in (torch.float32, torch.bfloat16):
    xz = torch.randn(B, 2 * D, L, device=dev).to(dt)
    dxz = torch.empty_like(xz)
    du = torch.randn(B, D, L, device=dev).to(dt)
    cw = torch.randn(D, W, device=dev)
    cb = torch.randn(D, device=dev)
    acc = torch.empty(D * W + D, device=dev)
    st = _lib.stream_ptr(dev)

    def run():
        rc = lib.simamba_causal_conv1d_bwd(xz.data_ptr(), cw.data_ptr(), cb.data_ptr(), du.data_ptr(), dxz.data_ptr(),
                                           acc.data_ptr(), acc[D * W:].data_ptr(), B, D, L, W, 1, _lib.dtype_code(dt),
                                           xz.stride(0), dxz.stride(0), st)
        assert rc == 0
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    byt = 3 * B * D * L * xz.element_size()
    print(f"{str(dt)[6:]:9s} conv1d_bwd {us:7.1f} us   {byt / us / 1e6:6.2f} TB/s of {byt / 1e6:.0f} MB")
