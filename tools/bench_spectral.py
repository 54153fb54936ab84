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

# rocSOLVER baseline: torch.linalg.eigh on the device = hipSOLVER/rocSOLVER syevd, batched and in the
# reference's per-sample loop (models/point_mamba.py:725-742), on the same Laplacians
from si_mamba_amd.spectral import create_graph_from_centers
for B, G in [(64, 128), (64, 64)]:
    c = unit_ball_centers(B, G, 0).to(dev)
    adj = create_graph_from_centers(c, 20, 10.0, True, False, True)
    A = (adj + adj.transpose(1, 2)) / 2
    Lrw = torch.eye(G, device=dev)[None] - (1.0 / (A.sum(2) + 1e-6))[:, :, None] * A
    for name, fn in [("batched", lambda: torch.linalg.eigh(Lrw)),
                     ("per-sample loop", lambda: [torch.linalg.eigh(Lrw[i]) for i in range(B)])]:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort()
        print(f"rocSOLVER eigh {name} B={B} G={G}: median {ts[2]:.2f} ms -> {B/ts[2]*1e3:.0f} matrices/s")
