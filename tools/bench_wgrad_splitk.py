"""Encoder weight gradients dW = dY^T X with M = B*G*n = 262144 rows: one plain GEMM (what autograd's mm backward
runs; the tuned table applies) against a split-K batched product + sum (tuning tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd.gemm_tuning import enable_tuned_gemms

dev = torch.device("cuda:0")
enable_tuned_gemms()

def timeit(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

M = 262144
for dt in (torch.float32, torch.bfloat16):
    for N, K in ((384, 512), (512, 256), (256, 128), (128, 3)):
        dy = torch.randn(M, N, device=dev, dtype=dt); x = torch.randn(M, K, device=dev, dtype=dt)
        ref = dy.double().t() @ x.double() if N * K < 1e5 else None
        t_plain = timeit(lambda: dy.t() @ x)
        line = f"{str(dt)[6:]:9s} N={N:3d} K={K:3d}  plain {t_plain:8.1f} us"
        for S in (32, 64, 128, 256):
            m = M // S
            def split():
                a = dy.view(S, m, N).transpose(1, 2)
                b = x.view(S, m, K)
                if dt == torch.float32:
                    return torch.bmm(a, b).sum(0)
                return torch.bmm(a, b, out_dtype=torch.float32).sum(0)
            try:
                t = timeit(split)
                line += f" | S={S}: {t:7.1f}"
            except Exception as e:
                line += f" | S={S}: {type(e).__name__}"
        if ref is not None:
            g = split()
            line += f" | err {((g.double() - ref).abs().max() / ref.abs().max()).item():.1e}"
        print(line, flush=True)
