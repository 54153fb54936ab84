"""in_proj at the model shape (B=64, L=1024, 384 -> 1536), bf16: the hand-written kernel (csrc/in_proj_bf16.hip) against
the library GEMM it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
if os.environ.get("SIMAMBA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SIMAMBA_LIB"])

dev = torch.device("cuda:0")
lib = _lib.load()
B, L, C, M = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 384, int(os.environ.get("M", 1536))
dt = torch.float32 if os.environ.get("DTYPE", "bf16") == "f32" else torch.bfloat16
if dt == torch.float32:
    from si_mamba_amd.gemm_tuning import enable_tuned_gemms
    enable_tuned_gemms()
x = torch.randn(B, L, C, device=dev).to(dt)
w = (torch.randn(M, C, device=dev) * C ** -0.5).to(dt)
xz = torch.empty(B, M, L, device=dev, dtype=dt)
st = _lib.stream_ptr(dev)


def hand():
    assert lib.simamba_in_proj_fwd(x.data_ptr(), w.data_ptr(), xz.data_ptr(), B, L, C, M, _lib.dtype_code(x.dtype), st) == 0


def library():
    return torch.bmm(w.unsqueeze(0).expand(B, -1, -1), x.transpose(1, 2))


for name, fn in (("hand in_proj kernel", hand), ("library GEMM", library)) * 2:
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    byt = (B * L * C + M * C + B * M * L) * x.element_size()
    print(f"{str(dt)[6:]} B={B} L={L} M={M} {name:22s} {us:7.1f} us   {byt / us / 1e6:5.2f} TB/s of {byt / 1e6:.0f} MB   "
          f"{2 * B * L * C * M / us * 1e-6:.0f} TF/s", flush=True)
