"""Kernel micro-benchmark for tuning (not the judged bench): scan fwd/bwd + conv + spectral at the
north-star micro-shape and the model-level shape.  SIMAMBA_LIB=<path> selects an alternative build."""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
if os.environ.get("SIMAMBA_LIB"):
    _lib.LIB_PATH = os.environ["SIMAMBA_LIB"]
from si_mamba_amd import selective_scan_fn
from si_mamba_amd.synthetic import scan_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="256x768x128,64x768x1024")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--modes", default="fwd,bwd")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--variant", type=int, default=0, help="forward kernel: 0 auto, 1 row-scan, 2 / 4 lanes per channel, 6 mixed 2 + 4")
args = ap.parse_args()
_lib._scan_variant[0] = args.variant
dev = torch.device("cuda:0")
dt = torch.float32 if args.dtype == "f32" else torch.bfloat16
res = {}
for shp in args.shapes.split(","):
    B, D, L = map(int, shp.split("x"))
    N = 16
    s = 4 if dt == torch.float32 else 2
    t = {k: (v.to(dev) if v is not None else None) for k, v in scan_inputs(B, D, L, N, seed=0).items()}
    for k in ("u", "delta", "z", "B", "C", "dout"):
        t[k] = t[k].to(dt)
    leaves = [t[k].requires_grad_(True) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")]
    for mode in args.modes.split(","):
        times = []
        for i in range(args.iters + 3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if mode == "fwd":
                with torch.no_grad():
                    a.record()
                    selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)
                    b.record()
            else:
                out = selective_scan_fn(*leaves[:6], z=leaves[6], delta_bias=leaves[7], delta_softplus=True)
                a.record()
                torch.autograd.grad(out, leaves, t["dout"])
                b.record()
            torch.cuda.synchronize()
            if i >= 3:
                times.append(a.elapsed_time(b))
        times.sort()
        med = times[len(times) // 2]
        nbytes = (4 * B * D * L * s + 2 * B * N * L * s) if mode == "fwd" else (7 * B * D * L * s + 2 * B * N * L * (s + 4))
        res[f"{shp}:{mode}"] = {"ms": round(med, 4), "min_ms": round(times[0], 4), "GBs": round(nbytes / med / 1e6, 1)}
        print(f"{os.environ.get('SIMAMBA_LIB', 'default'):40s} {shp:16s} {mode}: median {med*1e3:8.1f} us  min {times[0]*1e3:8.1f} us  {nbytes/med/1e6:7.1f} GB/s ({nbytes/med/1e6/80:.1f}% of 8 TB/s)", flush=True)
print(json.dumps(res))
