"""The mixer's weight gradients dW = sum_b X[b] @ Y[b] (mamba_inner._sum_bmm): the batched product + aten sum it runs
against the alternatives that would drop the (B, M, N) partials and the reduce launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd.gemm_tuning import enable_tuned_gemms

dev = torch.device("cuda:0")
enable_tuned_gemms()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


B, D, L, d = 64, 768, 1024, 384
for dt in (torch.float32, torch.bfloat16):
    # in_proj: dxz (B, 2D, L) x hidden (B, L, d); out_proj: dout^T (B, d, L) x y^T (B, L, D); x_proj: dx_dbl^T (B, 56, L) x x^T
    for name, M, N in (("in_proj", 2 * D, d), ("out_proj", d, D), ("x_proj", 56, D)):
        X = torch.randn(B, M, L, device=dev, dtype=dt)
        Y = torch.randn(B, L, N, device=dev, dtype=dt)
        t_bmm = timeit(lambda: torch.bmm(X, Y))
        t_sum = timeit(lambda: torch.bmm(X, Y).sum(0))

        def accumulate():                                   # one (M, N) buffer, beta = 1, a launch per sample
            out = torch.zeros(M, N, device=dev, dtype=dt)
            for b in range(B):
                out.addmm_(X[b], Y[b])
            return out

        def slabs(S=8):                                     # S-sample batched products accumulated by baddbmm (beta = 1)
            out = torch.zeros(S, M, N, device=dev, dtype=dt)
            for s in range(0, B, S):
                torch.baddbmm(out, X[s:s + S], Y[s:s + S], out=out)
            return out.sum(0)

        def one_gemm():                                     # K = B L in one product: needs a (M, B L) copy of X
            return X.transpose(0, 1).reshape(M, B * L) @ Y.reshape(B * L, N)

        print(f"{str(dt)[6:]:9s} {name:9s} bmm alone {t_bmm:7.1f} us | bmm + sum {t_sum:7.1f} | 64 x addmm_ {timeit(accumulate):7.1f}"
              f" | 8 x baddbmm(8) + sum {timeit(slabs):7.1f} | one K=B*L GEMM (+ operand copy) {timeit(one_gemm):7.1f}")
