"""Does an fp32 weight-gradient GEMM (matrix pipe / vector issue bound) overlap with the HBM-bound kernels of the same
backward (conv1d backward, add + LayerNorm backward) when issued on a second stream?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
from si_mamba_amd.gemm_tuning import enable_tuned_gemms

dev = torch.device("cuda:0")
enable_tuned_gemms()
lib = _lib.load()
B, D, L, d, W = 64, 768, 1024, 384, 4
dt = torch.float32
xz = torch.randn(B, 2 * D, L, device=dev)
dxz = torch.empty_like(xz)
du = torch.randn(B, D, L, device=dev)
cw, cb = torch.randn(D, W, device=dev), torch.randn(D, device=dev)
acc = torch.empty(D * W + D, device=dev)
dout = torch.randn(B, L, d, device=dev)
y = torch.randn(B, D, L, device=dev)
hidden = torch.randn(B, L, d, device=dev)
# add_ln_bwd operands
dn, dro, res = (torch.randn(B, L, d, device=dev) for _ in range(3))
mean, rstd = torch.randn(B * L, device=dev), torch.rand(B * L, device=dev) + 0.5
lw = torch.randn(d, device=dev)
dres = torch.empty(B, L, d, device=dev)
grid = lib.simamba_add_layer_norm_grid(B, L)
part = torch.empty(grid, 2, d, device=dev)


def mem_kernels(st):
    rc = lib.simamba_causal_conv1d_bwd(xz.data_ptr(), cw.data_ptr(), cb.data_ptr(), du.data_ptr(), dxz.data_ptr(),
                                       acc.data_ptr(), acc[D * W:].data_ptr(), B, D, L, W, 1, 0, xz.stride(0),
                                       dxz.stride(0), st)
    assert rc == 0
    rc = lib.simamba_add_layer_norm_bwd(dn.data_ptr(), dro.data_ptr(), res.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                        lw.data_ptr(), None, dres.data_ptr(), None, part.data_ptr(), B, L, d, 0, 0, st)
    assert rc == 0


def wgrad_out():
    return torch.bmm(dout.transpose(1, 2), y.transpose(1, 2)).sum(0)


def wgrad_in():
    return torch.bmm(dxz, hidden).sum(0)


main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def par(g):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        g()
    mem_kernels(main.cuda_stream)
    main.wait_stream(side)


print("conv1d_bwd + add_ln_bwd alone     %.1f us" % timeit(lambda: mem_kernels(main.cuda_stream)))
for name, g in (("out_proj wgrad", wgrad_out), ("in_proj wgrad", wgrad_in)):
    t = timeit(g)
    print("%-16s alone              %.1f us" % (name, t))
    print("  one stream                        %.1f us" % timeit(lambda: (g(), mem_kernels(main.cuda_stream))))
    print("  two streams                       %.1f us" % timeit(lambda: par(g)))
