import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from si_mamba_amd import spectral
from si_mamba_amd.synthetic import unit_ball_centers
dev = torch.device("cuda:0")
for B, G in [(64, 128), (128, 128), (256, 128), (64, 64), (512, 64)]:
    c = unit_ball_centers(B, G, 0).to(dev)
    for _ in range(3):
        spectral.spectral_order(c, 20, 10.0, 4, smallest=True, symmetric=True, self_loop=False, binary=True)
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); spectral.spectral_order(c, 20, 10.0, 4, smallest=True, symmetric=True, self_loop=False, binary=True); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"spectral_order B={B} G={G}: median {ts[5]:.3f} ms -> {B/ts[5]*1e3:.0f} matrices/s")
