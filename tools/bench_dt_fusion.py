"""conv + x_proj (+ dt_proj) -> scan forward -> scan backward of one mixer at the model shape, with delta materialised by
the xdt kernel (round 2) and with delta formed inside the scan kernels; kernels launched back to back in the order the
mixer launches them, each bracketed by events.  Usage: python tools/bench_dt_fusion.py [B] [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
D, L, N, R, S = 768, 1024, 16, 24, 56
for dtype in (torch.float32, torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    xz = rn(B, 2 * D, L).to(dtype)
    x_in, z = xz[:, :D], xz[:, D:]
    cw, cb = (0.5 * rn(D, 4)).contiguous(), rn(D)
    wx, wdt = (rn(S, D) * D ** -0.5).to(dtype), (rn(D, R) * R ** -0.5).to(dtype)
    A = -torch.exp(torch.log(torch.arange(1, 17, device=dev).float())[None].expand(D, -1) + 0.1 * rn(D, 16)).contiguous()
    Dp, bias = torch.ones(D, device=dev), torch.log(torch.expm1(torch.exp(torch.rand(D, device=dev) * 4.6 - 6.9)))
    dout = rn(B, D, L).to(dtype)
    x_conv, y, delta = (torch.empty(B, D, L, device=dev, dtype=dtype) for _ in range(3))
    du, dd = (torch.empty(B, D, L, device=dev, dtype=dtype) for _ in range(2))
    dxz = torch.empty_like(xz)
    x_dbl = torch.empty(B, L, S, device=dev, dtype=dtype)
    ck = torch.empty(lib.simamba_scan_ckpt_floats(B, D, L, N, 16), device=dev)
    acc = _lib.scan_bwd_accumulators(B, D, L, N, True, True, dev)
    st, code, xbs = _lib.stream_ptr(dev), _lib.dtype_code(dtype), xz.stride(0)
    Bv, Cv = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:]

    def xdt(fused):
        return lib.simamba_conv_xdt_proj_fwd(x_in.data_ptr(), cw.data_ptr(), cb.data_ptr(), wx.data_ptr(), wdt.data_ptr(),
                                             x_conv.data_ptr(), x_dbl.data_ptr(), None if fused else delta.data_ptr(),
                                             B, D, L, S, R, code, xbs, st)

    def fwd(fused):
        if fused:
            return lib.simamba_selective_scan_dt_fwd(x_conv.data_ptr(), x_dbl.data_ptr(), wdt.data_ptr(), A.data_ptr(),
                                                     Dp.data_ptr(), z.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                     ck.data_ptr(), None, B, D, L, N, R, code, xbs, 0, 0, 16, variant, st)
        return lib.simamba_selective_scan_fwd(x_conv.data_ptr(), delta.data_ptr(), A.data_ptr(), Bv.data_ptr(), Cv.data_ptr(),
                                              Dp.data_ptr(), z.data_ptr(), bias.data_ptr(), y.data_ptr(), ck.data_ptr(), None,
                                              B, D, L, N, code, 1, xbs, x_dbl.stride(0), 1, x_dbl.stride(1), 16, variant, st)

    def bwd(fused):
        if fused:
            return lib.simamba_selective_scan_dt_bwd(x_conv.data_ptr(), x_dbl.data_ptr(), wdt.data_ptr(), A.data_ptr(),
                                                     Dp.data_ptr(), z.data_ptr(), bias.data_ptr(), dout.data_ptr(),
                                                     ck.data_ptr(), du.data_ptr(), dd.data_ptr(), acc[0].data_ptr(),
                                                     acc[1].data_ptr(), acc[2].data_ptr(), acc[3].data_ptr(),
                                                     dxz[:, D:].data_ptr(), acc[4].data_ptr(), B, D, L, N, R, code, xbs,
                                                     dxz.stride(0), 0, 0, st)
        return lib.simamba_selective_scan_bwd(x_conv.data_ptr(), delta.data_ptr(), A.data_ptr(), Bv.data_ptr(), Cv.data_ptr(),
                                              Dp.data_ptr(), z.data_ptr(), bias.data_ptr(), dout.data_ptr(), ck.data_ptr(),
                                              du.data_ptr(), dd.data_ptr(), acc[0].data_ptr(), acc[1].data_ptr(),
                                              acc[2].data_ptr(), acc[3].data_ptr(), dxz[:, D:].data_ptr(),
                                              acc[4].data_ptr(), B, D, L, N, code, 1, xbs, dxz.stride(0), x_dbl.stride(0), 1,
                                              x_dbl.stride(1), 16, st)

    for fused in (False, True, False, True):
        tot = {"xdt": 0.0, "fwd": 0.0, "bwd": 0.0}
        n = 10
        for it in range(n + 2):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(); assert xdt(fused) == 0
            ev[1].record(); assert fwd(fused) == 0
            ev[2].record(); assert bwd(fused) == 0
            ev[3].record()
            torch.cuda.synchronize()
            if it >= 2:
                for i, k in enumerate(("xdt", "fwd", "bwd")):
                    tot[k] += ev[i].elapsed_time(ev[i + 1]) * 1e3 / n
        print(f"B={B} {str(dtype)[6:]:9s} {'in-kernel delta' if fused else 'materialised  '}  " +
              "  ".join(f"{k} {v:6.1f} us" for k, v in tot.items()) + f"  sum {sum(tot.values()):7.1f} us", flush=True)
