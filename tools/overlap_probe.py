"""Do a VALU-bound scan kernel and an MFMA-bound library GEMM overlap when issued on two HIP streams?  (diagnostic)

Backward of one mixer at the bench shape: scan bwd (VALU-bound; the sequential kernel holds 3 waves/SIMD at <= 168 VGPRs and
53 KB of LDS per workgroup, SIMAMBA_CKPT=128 selects the row-scan one: 2 waves at 241) next to the two weight-gradient
GEMMs that are off the critical path (out_proj wgrad, in_proj wgrad; fp32 MFMA at ~140 TF/s).  Prints the time of the two
run back to back on one stream and of the same work issued on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
if os.environ.get("SIMAMBA_LIB"):                                  # an A/B build from tools/build_alt.sh
    _lib.LIB_PATH = os.path.abspath(os.environ["SIMAMBA_LIB"])
from si_mamba_amd.gemm_tuning import enable_tuned_gemms
from si_mamba_amd.synthetic import scan_inputs

dev = torch.device("cuda:0")
enable_tuned_gemms()
B, D, L, N = 64, 768, 1024, 16
t = {k: v.to(dev) for k, v in scan_inputs(B, D, L, N, seed=0).items()}
lib = _lib.load()
out = torch.empty_like(t["u"])
CK = int(os.environ.get("SIMAMBA_CKPT", "0")) or lib.simamba_scan_ckpt_step(B, D, L, N, 0)   # 16: sequential backward
x_ckpt = torch.empty(lib.simamba_scan_ckpt_floats(B, D, L, N, CK), device=dev)
du, dd, dz = (torch.empty_like(t["u"]) for _ in range(3))
acc = _lib.scan_bwd_accumulators(B, D, L, N, True, True, dev)


def scan_fwd(st):
    assert lib.simamba_selective_scan_fwd(t["u"].data_ptr(), t["delta"].data_ptr(), t["A"].data_ptr(), t["B"].data_ptr(),
                                          t["C"].data_ptr(), t["D"].data_ptr(), t["z"].data_ptr(), t["delta_bias"].data_ptr(),
                                          out.data_ptr(), x_ckpt.data_ptr(), None, B, D, L, N, 0, 1, 0, 0, 0, 0, CK, 0, st) == 0


def scan_bwd(st):
    assert lib.simamba_selective_scan_bwd(t["u"].data_ptr(), t["delta"].data_ptr(), t["A"].data_ptr(), t["B"].data_ptr(),
                                          t["C"].data_ptr(), t["D"].data_ptr(), t["z"].data_ptr(), t["delta_bias"].data_ptr(),
                                          t["dout"].data_ptr(), x_ckpt.data_ptr(), du.data_ptr(), dd.data_ptr(),
                                          acc[0].data_ptr(), acc[1].data_ptr(), acc[2].data_ptr(), acc[3].data_ptr(),
                                          dz.data_ptr(), acc[4].data_ptr(), B, D, L, N, 0, 1, 0, 0, 0, 0, 0, CK, st) == 0


dxz = torch.randn(B, 2 * D, L, device=dev)
hidden = torch.randn(B, L, 384, device=dev)
dout = torch.randn(B, L, 384, device=dev)
y = torch.randn(B, D, L, device=dev)


def wgrads():
    a = torch.bmm(dxz, hidden).sum(0)                              # in_proj weight gradient
    b = torch.bmm(dout.transpose(1, 2), y.transpose(1, 2)).sum(0)  # out_proj weight gradient
    return a, b


main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
scan_fwd(main.cuda_stream)
torch.cuda.synchronize()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def seq():
    scan_bwd(main.cuda_stream)
    wgrads()


def par():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        wgrads()
    scan_bwd(main.cuda_stream)
    main.wait_stream(side)


print("scan bwd alone   %.1f us" % timeit(lambda: scan_bwd(main.cuda_stream)))
print("wgrad GEMMs alone %.1f us" % timeit(wgrads))
print("one stream        %.1f us" % timeit(seq))
print("two streams       %.1f us" % timeit(par))
print("two streams (scan first issued) %.1f us" % timeit(lambda: (scan_bwd(main.cuda_stream), side.wait_stream(main), None) and None or par()))
