"""Diagnostic: which orientation of the mixer's skinny projections hipBLASLt runs well (fp32, B=64, L=1024)."""
import torch
dev = torch.device("cuda:0")
B, L, D, S, R, d = 64, 1024, 768, 56, 24, 384
x = torch.randn(B, D, L, device=dev)
Wx = torch.randn(S, D, device=dev)
Wdt = torch.randn(D, R, device=dev)
xd_t = torch.randn(B, S, L, device=dev)       # x_dbl^T (B, S, L)
xd = torch.randn(B, L, S, device=dev)         # x_dbl   (B, L, S)
dd = torch.randn(B, D, L, device=dev)         # ddelta / du


def t(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:70s} {a.elapsed_time(b)/n*1e3:8.1f} us")

ex = lambda W: W.unsqueeze(0).expand(B, -1, -1)
t("x_proj  cur : bmm(x^T (B,L,D), Wx^T) -> (B,L,S)", lambda: torch.bmm(x.transpose(1, 2), ex(Wx.t())))
t("x_proj  alt : bmm(Wx (B,S,D), x (B,D,L)) -> (B,S,L)", lambda: torch.bmm(ex(Wx), x))
t("dt_proj cur : bmm(Wdt, xd[:,:,:R]^T) -> (B,D,L)", lambda: torch.bmm(ex(Wdt), xd[:, :, :R].transpose(1, 2)))
t("dt_proj alt : bmm(Wdt, xd_t[:,:R,:]) -> (B,D,L)", lambda: torch.bmm(ex(Wdt), xd_t[:, :R, :]))
t("ddt     cur : bmm(dd^T (B,L,D), Wdt) -> (B,L,R)", lambda: torch.bmm(dd.transpose(1, 2), ex(Wdt)))
t("ddt     alt : bmm(Wdt^T (B,R,D), dd (B,D,L)) -> (B,R,L)", lambda: torch.bmm(ex(Wdt.t()), dd))
t("d_dt_w  cur : bmm(dd (B,D,L), xd[:,:,:R]).sum(0)", lambda: torch.bmm(dd, xd[:, :, :R]).sum(0))
t("d_dt_w  alt : bmm(dd (B,D,L), xd_t[:,:R,:]^T).sum(0)", lambda: torch.bmm(dd, xd_t[:, :R, :].transpose(1, 2)).sum(0))
t("d_x_w   cur : bmm(xd^T (B,S,L), x^T (B,L,D)).sum(0)", lambda: torch.bmm(xd.transpose(1, 2), x.transpose(1, 2)).sum(0))
t("d_x_w   alt : bmm(xd_t (B,S,L), x^T (B,L,D)).sum(0)", lambda: torch.bmm(xd_t, x.transpose(1, 2)).sum(0))
du = dd.clone()
t("du+=    cur : baddbmm(du, Wx^T (B,D,S), xd^T (B,S,L))", lambda: torch.baddbmm(du, ex(Wx.t()), xd.transpose(1, 2), out=du))
t("du+=    alt : baddbmm(du, Wx^T (B,D,S), xd_t (B,S,L))", lambda: torch.baddbmm(du, ex(Wx.t()), xd_t, out=du))
# one big GEMM alternatives for weight grads: (D, B*L) x (B*L, R) needs a (B*L) stride -> not expressible; einsum:
t("d_dt_w  ein : einsum('bdl,brl->dr', dd, xd_t[:,:R])", lambda: torch.einsum("bdl,brl->dr", dd, xd_t[:, :R]))
t("d_x_w   ein : einsum('bsl,bdl->sd', xd_t, x)", lambda: torch.einsum("bsl,bdl->sd", xd_t, x))
