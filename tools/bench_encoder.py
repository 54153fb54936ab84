"""Encoder streaming kernels micro-benchmark (tuning tool): bn_relu fwd/bwd and group max at the bench shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from si_mamba_amd.encoder_ops import bn_relu_fn, group_max_fn
dev = torch.device("cuda:0")

def timeit(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]

for rows, C, group in [(262144, 128, 0), (262144, 512, 32)]:
    x = torch.randn(rows, C, device=dev, requires_grad=True)
    g = torch.randn(rows // group, C, device=dev) if group else None
    dy = torch.randn(rows, C, device=dev)
    bn = nn.BatchNorm1d(C).to(dev)
    mb = rows * C * 4 / 1e6
    t = timeit(lambda: bn_relu_fn(x.detach(), bn, gterm=g, group=group))
    print(f"bn_relu fwd ({rows},{C}): {t*1e3:7.1f} us  {3*mb/t/1e3:6.2f} TB/s (3 passes)")
    y = bn_relu_fn(x, bn, gterm=g, group=group)
    t = timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
    print(f"bn_relu bwd ({rows},{C}): {t*1e3:7.1f} us  {5*mb/t/1e3:6.2f} TB/s (5 passes)")
    ref = lambda: torch.relu(bn(x.detach() if g is None else x.detach() + g.repeat_interleave(group, 0)))
    print(f"  torch bn+relu fwd: {timeit(ref)*1e3:7.1f} us")
for groups, n, C in [(8192, 32, 256), (8192, 32, 384)]:
    x = torch.randn(groups, n, C, device=dev, requires_grad=True)
    mb = groups * n * C * 4 / 1e6
    t = timeit(lambda: group_max_fn(x.detach()))
    print(f"group_max fwd ({groups},{n},{C}): {t*1e3:7.1f} us  {mb/t/1e3:6.2f} TB/s")
    y = group_max_fn(x); dy = torch.randn_like(y)
    t = timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
    print(f"group_max bwd: {t*1e3:7.1f} us  {mb/t/1e3:6.2f} TB/s;  torch max fwd {timeit(lambda: x.detach().max(dim=1)[0])*1e3:.1f} us")

# ---- segmentation / pre-training head kernels ---------------------------------------------------------------
from si_mamba_amd.interp import three_interpolate, three_nn
from si_mamba_amd.mae import chamfer_distance
from si_mamba_amd.grouping import knn_group, sample_farthest_points
from si_mamba_amd.synthetic import make_clouds
B, N, S, C = 16, 2048, 256, 1152
pts = make_clouds(B, N, 0, dev)
centres = pts[:, :S].contiguous()
feats = torch.randn(B, S, C, device=dev, requires_grad=True)
t = timeit(lambda: three_nn(pts, centres))
print(f"three_nn ({B},{N}) x {S} centres: {t*1e3:7.1f} us")
idx, w = three_nn(pts, centres)
t = timeit(lambda: three_interpolate(feats.detach(), idx, w))
mb = B * N * C * 4 / 1e6
print(f"three_interpolate fwd -> ({B},{N},{C}): {t*1e3:7.1f} us  {mb/t/1e3:5.2f} TB/s of output bytes")
out = three_interpolate(feats, idx, w); dout = torch.randn_like(out)
t = timeit(lambda: torch.autograd.grad(out, feats, dout, retain_graph=True))
print(f"three_interpolate bwd (gather): {t*1e3:7.1f} us  {mb/t/1e3:5.2f} TB/s of dout bytes")
pairs = 64 * 304
pred = torch.randn(pairs, 32, 3, device=dev, requires_grad=True); gt = torch.randn(pairs, 32, 3, device=dev)
t = timeit(lambda: chamfer_distance(pred.detach(), gt))
print(f"chamfer fwd {pairs} pairs of 32x32: {t*1e3:7.1f} us")
d = chamfer_distance(pred, gt)
t = timeit(lambda: torch.autograd.grad(d.sum(), pred, retain_graph=True))
print(f"chamfer bwd: {t*1e3:7.1f} us")
cl = make_clouds(64, 1024, 1, dev)
cen, _ = sample_farthest_points(cl, 128)
t = timeit(lambda: knn_group(cen, cl, 32))
ref = timeit(lambda: torch.cdist(cen, cl).topk(32, dim=-1, largest=False, sorted=False)[1])
print(f"knn_group (64,1024)->(64,128,32): {t*1e3:7.1f} us   (cdist + topk: {ref*1e3:.1f} us)")
