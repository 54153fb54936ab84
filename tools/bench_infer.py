"""Inference throughput of the PointMamba classifier, eager vs one-hipGraph replay (small batches are
launch-bound in eager mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd.graphed import GraphedForward
from si_mamba_amd.point_mamba import PointMamba, default_config
from si_mamba_amd.synthetic import make_clouds
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointMamba(default_config()).to(dev).eval()
for B in (1, 8, 32, 64):
    x = make_clouds(B, 1024, 0, dev)
    with torch.no_grad():
        for _ in range(3): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m(x)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
    g = GraphedForward(m, x)
    for _ in range(3): g(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g(x)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    print(f"B={B:3d}: eager {te*1e3:7.2f} ms ({B/te:8.1f} clouds/s)   hipGraph {tg*1e3:7.2f} ms ({B/tg:8.1f} clouds/s)", flush=True)
