"""Development check of the sequential scan backward (scan_bwd_seq.hip): parity against the oracle and the row-scan
backward on small shapes, timing at the model / micro shapes.  Usage: python tools/dev_bwd_seq.py [--time]"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import scan_ref                        # noqa: E402
from oracle.gen_golden import scan_inputs          # noqa: E402
from si_mamba_amd import _lib, selective_scan_fn   # noqa: E402

if os.environ.get("SIMAMBA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SIMAMBA_LIB"])
dev = torch.device("cuda:0")
QUICK = "--quick" in sys.argv          # timing of the model shape only, fp32


def nerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()


def run(inp, dtype, step, variant=_lib.SCAN_AUTO, token_major=False):
    act = ("u", "delta", "z", "B", "C")
    t = {}
    for k, v in inp.items():
        if v is None or k == "dout":
            t[k] = v
            continue
        x = v.to(dev)
        if k in act:
            x = x.to(dtype)
        if token_major and k in ("B", "C"):
            x = x.transpose(1, 2).contiguous().transpose(1, 2).detach()
        t[k] = x.requires_grad_(True)
    with _lib.scan_ckpt(step), _lib.scan_variant(variant):
        out = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"],
                                delta_softplus=True)
        out.backward(inp["dout"].to(dev).to(dtype))
    g = {k: t[k].grad for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias") if t.get(k) is not None}
    return out, g


def oracle(inp, dtype):
    act = ("u", "delta", "z", "B", "C")
    t = {}
    for k, v in inp.items():
        if v is None or k == "dout":
            t[k] = v
            continue
        x = v.clone().to(dtype).float() if k in act else v.clone()
        t[k] = x.requires_grad_(True)
    out = scan_ref.selective_scan_ref(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"],
                                      delta_softplus=True)
    out.backward(inp["dout"].to(dtype).float())
    return out, {k: t[k].grad for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias") if t.get(k) is not None}


bad = 0
for (b, d, L, N, kw) in [] if QUICK else [(2, 64, 32, 16, {}), (2, 64, 64, 16, {}), (1, 128, 48, 16, {}), (2, 64, 208, 16, {}),
                         (1, 64, 16, 16, {}), (1, 64, 4, 16, {}), (2, 192, 512, 16, {}), (2, 64, 100, 16, {}),
                         (2, 64, 96, 16, dict(with_z=False)), (2, 64, 96, 16, dict(with_D=False, with_bias=False))]:
    for dtype in (torch.float32, torch.bfloat16):
        if dtype == torch.bfloat16 and L % 8:
            continue
        for tm in (False, True):
            inp = scan_inputs(b, d, L, N, seed=L + d, **kw)
            wout, wg = oracle(inp, dtype)
            out, g = run(inp, dtype, _lib.CKPT_SEQ, token_major=tm)
            tol = 1e-3 if dtype == torch.float32 else 1e-2
            errs = {k: nerr(g[k], wg[k]) for k in wg}
            errs["out"] = nerr(out, wout)
            worst = max(errs.values())
            flag = "" if worst < tol else "  <-- FAIL"
            bad += worst >= tol
            print((b, d, L, N), kw, str(dtype)[6:], "tok" if tm else "time", " ".join(f"{k}:{v:.1e}" for k, v in errs.items()), flag)
print("FAILS:", bad)

if "--time" in sys.argv:
    lib = _lib.load()
    for (B, D, L) in [(64, 768, 1024), (256, 768, 128), (128, 768, 1024), (64, 768, 512), (64, 768, 208), (16, 768, 1024)][:1 if QUICK else 9]:
        for dtype in (torch.float32, torch.bfloat16)[:1 if QUICK else 2]:
            N = 16
            g = torch.Generator(device="cuda").manual_seed(0)
            def rn(*s):
                return torch.randn(*s, device=dev, generator=g)
            u, z, dout = (rn(B, D, L).to(dtype) for _ in range(3))
            delta = (0.5 * rn(B, D, L)).to(dtype)
            xdbl = rn(B, L, 56).to(dtype)
            Bm, Cm = xdbl[:, :, 24:40], xdbl[:, :, 40:]
            A = -torch.exp(torch.log(torch.arange(1, 17, device=dev).float())[None].expand(D, -1) + 0.1 * rn(D, 16)).contiguous()
            Dp = torch.ones(D, device=dev)
            bias = torch.log(torch.expm1(torch.exp(torch.rand(D, device=dev) * 4.6 - 6.9)))
            out = torch.empty_like(u)
            du, dd, dz = (torch.empty_like(u) for _ in range(3))
            acc = _lib.scan_bwd_accumulators(B, D, L, N, True, True, dev)
            st = _lib.stream_ptr(dev)
            code = _lib.dtype_code(dtype)
            res = {}
            for step in (_lib.CKPT_ROW, _lib.CKPT_SEQ):
                n = lib.simamba_scan_ckpt_floats(B, D, L, N, step)
                ck = torch.empty(n, device=dev) if n else None
                def fwd():
                    return lib.simamba_selective_scan_fwd(u.data_ptr(), delta.data_ptr(), A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(),
                        Dp.data_ptr(), z.data_ptr(), bias.data_ptr(), out.data_ptr(), _lib.ptr(ck), None, B, D, L, N, code, 1,
                        0, xdbl.stride(0), 1, xdbl.stride(1), step, 0, st)
                def bwd():
                    return lib.simamba_selective_scan_bwd(u.data_ptr(), delta.data_ptr(), A.data_ptr(), Bm.data_ptr(), Cm.data_ptr(),
                        Dp.data_ptr(), z.data_ptr(), bias.data_ptr(), dout.data_ptr(), _lib.ptr(ck), du.data_ptr(), dd.data_ptr(),
                        acc[0].data_ptr(), acc[1].data_ptr(), acc[2].data_ptr(), acc[3].data_ptr(), dz.data_ptr(), acc[4].data_ptr(),
                        B, D, L, N, code, 1, 0, 0, xdbl.stride(0), 1, xdbl.stride(1), step, st)
                for name, fn in (("fwd", fwd), ("bwd", bwd)):
                    for _ in range(3):
                        rc = fn()
                        assert rc == 0, (name, step, rc)
                    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(20):
                        fn()
                    e.record()
                    torch.cuda.synchronize()
                    res[(name, step)] = a.elapsed_time(e) / 20 * 1e3
                res[("grads", step)] = [x.clone() for x in (du, dd, dz, acc[0], acc[1], acc[2], acc[3], acc[4])]
            diff = max(nerr(a, b) for a, b in zip(res[("grads", 16)], res[("grads", 128)]))
            print(os.environ.get("SIMAMBA_LIB", "default"), (B, D, L), str(dtype)[6:], " ".join(f"{k[0]}@{k[1]}:{v:.0f}us" for k, v in res.items() if k[0] != "grads"),
                  f"seq-vs-row max nerr {diff:.1e}", flush=True)
