"""The forward-scan variants timed one call at a time between other kernels, as inside the model (tuning tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
if os.environ.get("SIMAMBA_LIB"):
    _lib.LIB_PATH = os.environ["SIMAMBA_LIB"]
from si_mamba_amd import selective_scan_fn
from si_mamba_amd.synthetic import scan_inputs

dev = torch.device("cuda:0")
B, D, L, N, R = 64, 768, 1024, 16, 24
t = {k: (v.to(dev) if v is not None else None) for k, v in scan_inputs(B, D, L, N, seed=0).items()}
a_mat = torch.randn(8192, 1024, device=dev); b_mat = torch.randn(1024, 4096, device=dev)
big = torch.empty(64 * 1536 * 1024, device=dev)

def scan(grad):
    if grad:
        u = t["u"].detach().requires_grad_(True)
        return selective_scan_fn(u, t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)
    with torch.no_grad():
        return selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)

hid = torch.randn(B * L, 384, device=dev); wz = torch.randn(384, D, device=dev) / 20
zbuf = torch.empty(B * L, D, device=dev)

def gemm_z():        # the z half of in_proj as a GEMM of its own between the producer of u / delta and the scan
    torch.mm(hid, wz, out=zbuf)

contexts = {
    "back to back": lambda: None,
    "after a GEMM": lambda: torch.mm(a_mat, b_mat),
    "after a 400 MB fill": lambda: big.fill_(1.0),
    "after GEMM + fill": lambda: (torch.mm(a_mat, b_mat), big.fill_(1.0)),
    "fill, then z GEMM": lambda: (big.fill_(1.0), gemm_z()),
    "after a 400 MB read": lambda: big.sum(),
    "after a 200 MB fill": lambda: big[:50 * 1024 * 1024].fill_(1.0),
    "after a 100 MB fill": lambda: big[:25 * 1024 * 1024].fill_(1.0),
}
for grad in (False, True):
    for cname, pre in contexts.items():
        for variant in (2, 6, 2, 6):
            _lib._scan_variant[0] = variant
            for _ in range(3):
                pre(); scan(grad)
            tot = 0.0
            n = 20
            evs = []
            for _ in range(n):
                pre()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); scan(grad); b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) for x, y in evs)
            print(f"grad={grad!s:5s} {cname:20s} variant {variant}: median {ts[n // 2] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f}", flush=True)
