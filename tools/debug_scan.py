import torch, sys
sys.path.insert(0, '.')
from oracle import scan_ref
from si_mamba_amd.synthetic import scan_inputs
from si_mamba_amd import selective_scan_fn
dev = torch.device('cuda:0')
def run(b, d, L, N, seed=0):
    inp = scan_inputs(b, d, L, N, seed=seed)
    keys = ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")
    t = {k: inp[k].to(dev).requires_grad_(True) for k in keys}
    r = {k: inp[k].clone().requires_grad_(True) for k in keys}
    out = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)
    out.backward(inp["dout"].to(dev))
    want = scan_ref.selective_scan_ref(r["u"], r["delta"], r["A"], r["B"], r["C"], r["D"], r["z"], r["delta_bias"], True)
    want.backward(inp["dout"])
    print(f"--- shape {(b,d,L,N)} out err {(out.cpu()-want).abs().max().item():.3e}")
    for k in keys:
        g, w = t[k].grad.cpu(), r[k].grad
        print(f"  d{k:11s} maxerr {(g-w).abs().max().item():.3e}  max|want| {w.abs().max().item():.3e}")
    return t, r
t, r = run(1, 4, 16, 1)
print("dB got ", t["B"].grad.cpu().flatten()[:16])
print("dB want", r["B"].grad.flatten()[:16])
print("dC got ", t["C"].grad.cpu().flatten()[:16])
print("dC want", r["C"].grad.flatten()[:16])
t, r = run(1, 1, 16, 1)
print("dB got ", t["B"].grad.cpu().flatten()[:16])
print("dB want", r["B"].grad.flatten()[:16])
t, r = run(1, 16, 128, 2)
print("dB got ", t["B"].grad.cpu()[0, 1, :16])
print("dB want", r["B"].grad[0, 1, :16])
run(2, 64, 64, 16)
run(2, 48, 300, 16)
