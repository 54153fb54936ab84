"""CPU baselines of BASELINE.md section 3 (C1-C4), timed on the host cores of the box the GPU numbers come from.
Uses the oracle (tests/bench-only code).  Writes one JSON object to stdout."""
import json, os, sys, time, platform
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import scan_ref, spectral_ref
from si_mamba_amd.synthetic import scan_inputs, unit_ball_centers


def med(fn, n=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


out = {"host": {"nproc": os.cpu_count(), "cpu": cpu_model(), "torch_threads_default": torch.get_num_threads()}}
for threads in (min(16, os.cpu_count()), 1):
    torch.set_num_threads(threads)
    rec = {}
    # C1: selective_scan_ref fwd and fwd+bwd
    for shape in ((2, 256, 64, 16), (256, 768, 128, 16)):
        if threads == 1 and shape[0] == 256:
            continue
        B, D, L, N = shape
        inp = scan_inputs(B, D, L, N, seed=0)
        args = (inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], inp["z"], inp["delta_bias"])
        nbytes = 4 * B * D * L * 4 + 2 * B * N * L * 4
        with torch.no_grad():
            t = med(lambda: scan_ref.selective_scan_ref(*args, delta_softplus=True), n=3)
        leaves = [a.clone().requires_grad_(True) for a in args]

        def fb():
            o = scan_ref.selective_scan_ref(*leaves, delta_softplus=True)
            torch.autograd.grad(o, leaves, inp["dout"])
        t2 = med(fb, n=3)
        rec[f"C1 scan_ref {shape}"] = {"fwd_ms": round(t * 1e3, 2), "fwd_GBs": round(nbytes / t / 1e9, 3),
                                       "fwd_bwd_ms": round(t2 * 1e3, 2)}
    # C2: one Block (LayerNorm + mixer) fwd+bwd, B=2, L=64, d=128
    torch.manual_seed(0)
    mix = scan_ref.MambaRef(128)
    ln = torch.nn.LayerNorm(128)
    h = torch.randn(2, 64, 128, requires_grad=True)
    rec["C2 block (2,64,128) fwd+bwd ms"] = round(med(lambda: mix(ln(h)).sum().backward(), n=5) * 1e3, 2)
    # C3: eigen path, per-sample loop (reference style) and batched
    for B, G in ((64, 64), (128, 128), (128, 64)):
        c = unit_ball_centers(B, G, 0)
        adj = spectral_ref.create_graph_from_feature_space(c, 20, 10.0, True, False, True)
        t = med(lambda: spectral_ref.calc_top_k_eigenvalues_eigenvectors(adj, 4, True), n=3)
        Lm = spectral_ref.rw_laplacian(adj)
        t2 = med(lambda: torch.linalg.eigh(Lm), n=3)
        rec[f"C3 eigh (B={B},G={G})"] = {"loop_matrices_per_s": round(B / t, 1), "batched_matrices_per_s": round(B / t2, 1)}
    out[f"threads={threads}"] = rec
print(json.dumps(out, indent=1))
