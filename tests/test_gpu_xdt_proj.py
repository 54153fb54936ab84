"""GPU parity of the fused x_proj -> dt_proj MFMA kernel (csrc/xdt_proj.hip) through the C ABI:
against the oracle's two matrix products in float64 (the kernel is an exact-fp32 MFMA chain: 1e-3 is the
north-star bar, the observed error is at the fp32 rounding level), and against the library-GEMM route of the mixer."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


# the last three walk several tiles per workgroup (> 512 tiles): D = 64 has two steps per tile (the four-step lookahead
# crosses two tile seams), D = 192 leaves a wave delta blocks past the in-loop slots, L = 616 has a ragged last tile
@pytest.mark.parametrize("B,D,L,N,R", [(2, 768, 1024, 16, 24), (3, 256, 64, 16, 8), (1, 768, 208, 16, 24),
                                       (2, 128, 36, 16, 8), (64, 768, 1024, 16, 24), (1, 64, 4, 16, 4),
                                       (40, 64, 1024, 16, 4), (60, 192, 640, 16, 12), (130, 128, 616, 16, 8)])
def test_xdt_proj_matches_float64_products(B, D, L, N, R, device):
    from si_mamba_amd.mamba_inner import xdt_proj_fused_ok, xdt_proj_fwd
    S = R + 2 * N
    g = torch.Generator().manual_seed(D + L)
    x = torch.randn(B, D, L, generator=g)
    wx = torch.randn(S, D, generator=g) / D ** 0.5
    wdt = torch.randn(D, R, generator=g) / R ** 0.5
    xd, wxd, wdd = x.to(device), wx.to(device), wdt.to(device)
    assert xdt_proj_fused_ok(xd, wxd, wdd)
    x_dbl, delta = xdt_proj_fwd(xd, wxd, wdd)
    sel = slice(None) if B <= 3 else torch.tensor([0, B // 2, B - 1])          # float64 oracle on a few samples
    want_dbl = torch.einsum("sd,bdt->bts", wx.double(), x[sel].double())
    want_delta = torch.einsum("dr,btr->bdt", wdt.double(), want_dbl[:, :, :R])
    assert nerr(x_dbl[sel], want_dbl) < 1e-5
    assert nerr(delta[sel], want_delta) < 1e-5
    # and bit-for-bit deterministic
    x_dbl2, delta2 = xdt_proj_fwd(xd, wxd, wdd)
    assert torch.equal(x_dbl, x_dbl2) and torch.equal(delta, delta2)


def test_xdt_proj_reads_a_batch_strided_input(device):
    from si_mamba_amd.mamba_inner import xdt_proj_fwd
    g = torch.Generator().manual_seed(3)
    big = torch.randn(3, 2 * 256, 128, generator=g).to(device)
    x = big[:, :256]                                                           # half of an in_proj-style (B, 2D, L)
    wx, wdt = torch.randn(40, 256, generator=g).to(device), torch.randn(256, 8, generator=g).to(device)
    a = xdt_proj_fwd(x, wx, wdt)
    b = xdt_proj_fwd(x.contiguous(), wx, wdt)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_mixer_uses_the_fused_kernel_and_matches_the_library_route(device, monkeypatch):
    """The fp32 mixer forward with the fused kernel against the same mixer forced onto the two library GEMMs:
    outputs and every gradient (the backward is shared) agree at the fp32 rounding level."""
    from si_mamba_amd import Mamba, mamba_inner
    torch.manual_seed(0)
    m = Mamba(384).to(device)
    h = torch.randn(2, 256, 384, device=device)
    calls = []
    real = mamba_inner.xdt_proj_fwd
    monkeypatch.setattr(mamba_inner, "xdt_proj_fwd", lambda *a, **k: (calls.append(k), real(*a, **k))[1])
    h1 = h.clone().requires_grad_(True)
    o1 = m(h1)
    o1.sum().backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert calls and calls[0].get("conv") is not None, "fp32 mixer forward did not take the fused conv + x_proj + dt_proj kernel"
    m.zero_grad(set_to_none=True)
    monkeypatch.setattr(mamba_inner, "xdt_proj_fused_ok", lambda *a, **k: False)
    h2 = h.clone().requires_grad_(True)
    o2 = m(h2)
    o2.sum().backward()
    assert nerr(o1, o2) < 1e-5 and nerr(h1.grad, h2.grad) < 1e-5
    for k, p in m.named_parameters():
        assert nerr(g1[k], p.grad) < 1e-4, k


@pytest.mark.parametrize("B,D,L", [(2, 768, 1024), (3, 256, 68), (1, 64, 4), (64, 768, 1024), (40, 64, 1024),
                                   (60, 192, 616)])
@pytest.mark.parametrize("bias", [True, False])
def test_conv_fused_into_the_staging_matches_the_conv_kernel(B, D, L, bias, device):
    """simamba_conv_xdt_proj_fwd: the conv output written as a by-product equals the stand-alone conv kernel's bit for
    bit (same fmaf chain), x_dbl / delta equal the unfused kernel's on that conv output, and everything matches the
    oracle's conv + float64 products on a few samples."""
    from oracle import scan_ref
    from si_mamba_amd import causal_conv1d_fn
    from si_mamba_amd.mamba_inner import xdt_proj_fwd
    N, R = 16, (24 if D == 768 else 8)
    S = R + 2 * N
    g = torch.Generator().manual_seed(L)
    xz = torch.randn(B, 2 * D, L, generator=g).to(device)
    x_in = xz[:, :D]
    cw = (torch.randn(D, 4, generator=g) * 0.5).to(device)
    cb = torch.randn(D, generator=g).to(device) if bias else None
    wx = (torch.randn(S, D, generator=g) / D ** 0.5).to(device)
    wdt = (torch.randn(D, R, generator=g) / R ** 0.5).to(device)
    x_conv = torch.empty(B, D, L, device=device)
    x_dbl, delta = xdt_proj_fwd(x_in, wx, wdt, conv=(cw, cb, x_conv))
    want_conv = causal_conv1d_fn(x_in, cw, cb, "silu")
    assert torch.equal(x_conv, want_conv)
    x_dbl2, delta2 = xdt_proj_fwd(want_conv, wx, wdt)
    assert torch.equal(x_dbl, x_dbl2) and torch.equal(delta, delta2)
    sel = [0, B - 1]
    oc = scan_ref.causal_conv1d_ref(x_in[sel].cpu(), cw.cpu(), None if cb is None else cb.cpu(), "silu")
    assert nerr(x_conv[sel], oc) < 1e-5
    want_dbl = torch.einsum("sd,bdt->bts", wx.cpu().double(), oc.double())
    assert nerr(x_dbl[sel], want_dbl) < 1e-5
    assert nerr(delta[sel], torch.einsum("dr,btr->bdt", wdt.cpu().double(), want_dbl[:, :, :R])) < 1e-5


# ---- bf16 form (v_mfma_f32_32x32x16_bf16; the reference's autocast roundings) ---------------------------------------
@pytest.mark.parametrize("B,D,L,N,R", [(2, 768, 1024, 16, 24), (3, 256, 72, 16, 8), (1, 768, 208, 16, 24),
                                       (1, 64, 8, 16, 4), (64, 768, 1024, 16, 24), (40, 64, 1024, 16, 4),
                                       (60, 192, 640, 16, 12), (130, 128, 616, 16, 8)])
def test_xdt_proj_bf16_matches_rounded_products(B, D, L, N, R, device):
    """bf16 operands: x_dbl is the float64 product of the bf16 values rounded once to bf16, delta the product of Wdt with
    those ROUNDED dt rows, rounded once (what x_proj followed by dt_proj give under autocast).  1e-2 is the north-star
    bar; the observed error is one bf16 rounding of values that sit on a rounding boundary."""
    from si_mamba_amd.mamba_inner import xdt_proj_fused_ok, xdt_proj_fwd
    S = R + 2 * N
    g = torch.Generator().manual_seed(D + L + 1)
    x = torch.randn(B, D, L, generator=g).bfloat16()
    wx = (torch.randn(S, D, generator=g) / D ** 0.5).bfloat16()
    wdt = (torch.randn(D, R, generator=g) / R ** 0.5).bfloat16()
    xd, wxd, wdd = x.to(device), wx.to(device), wdt.to(device)
    assert xdt_proj_fused_ok(xd, wxd, wdd)
    x_dbl, delta = xdt_proj_fwd(xd, wxd, wdd)
    assert x_dbl.dtype == torch.bfloat16 and delta.dtype == torch.bfloat16
    sel = slice(None) if B <= 3 else torch.tensor([0, B // 2, B - 1])
    want_dbl = torch.einsum("sd,bdt->bts", wx.double(), x[sel].double())
    assert nerr(x_dbl[sel], want_dbl) < 1e-2
    # delta from the kernel's own (rounded) dt rows: isolates the second product from the first one's rounding
    want_delta = torch.einsum("dr,btr->bdt", wdt.double(), x_dbl[sel][:, :, :R].double().cpu())
    assert nerr(delta[sel], want_delta) < 1e-2
    # mean error well below one bf16 ulp of the scale: no systematic loss (a dropped K block would show here)
    assert (x_dbl[sel].double().cpu() - want_dbl).abs().mean() < 4e-3 * want_dbl.abs().mean().clamp_min(1e-3)
    assert (delta[sel].double().cpu() - want_delta).abs().mean() < 4e-3 * want_delta.abs().mean().clamp_min(1e-3)
    x_dbl2, delta2 = xdt_proj_fwd(xd, wxd, wdd)
    assert torch.equal(x_dbl, x_dbl2) and torch.equal(delta, delta2)


@pytest.mark.parametrize("B,D,L", [(2, 768, 1024), (3, 256, 72), (1, 64, 8), (64, 768, 1024), (40, 64, 1024),
                                   (60, 192, 616)])
@pytest.mark.parametrize("bias", [True, False])
def test_conv_fused_bf16_matches_the_conv_kernel(B, D, L, bias, device):
    """bf16 conv-fused form: x_conv bit-identical to the stand-alone bf16 conv kernel (same fp32 fmaf chain, one rounding),
    x_dbl / delta bit-identical to the unfused bf16 kernel run on that conv output."""
    from si_mamba_amd import causal_conv1d_fn
    from si_mamba_amd.mamba_inner import xdt_proj_fwd
    N, R = 16, (24 if D == 768 else 8)
    S = R + 2 * N
    g = torch.Generator().manual_seed(L + 3)
    xz = torch.randn(B, 2 * D, L, generator=g).bfloat16().to(device)
    x_in = xz[:, :D]
    cw = (torch.randn(D, 4, generator=g) * 0.5).to(device)
    cb = torch.randn(D, generator=g).to(device) if bias else None
    wx = (torch.randn(S, D, generator=g) / D ** 0.5).bfloat16().to(device)
    wdt = (torch.randn(D, R, generator=g) / R ** 0.5).bfloat16().to(device)
    x_conv = torch.empty(B, D, L, device=device, dtype=torch.bfloat16)
    x_dbl, delta = xdt_proj_fwd(x_in, wx, wdt, conv=(cw, cb, x_conv))
    want_conv = causal_conv1d_fn(x_in, cw, cb, "silu")
    assert want_conv.dtype == torch.bfloat16 and torch.equal(x_conv, want_conv)
    x_dbl2, delta2 = xdt_proj_fwd(want_conv, wx, wdt)
    assert torch.equal(x_dbl, x_dbl2) and torch.equal(delta, delta2)


def test_bf16_mixer_uses_the_fused_kernel(device, monkeypatch):
    from si_mamba_amd import Mamba, mamba_inner
    torch.manual_seed(0)
    m = Mamba(384).to(device)
    h = torch.randn(2, 256, 384, device=device)
    calls = []
    real = mamba_inner.xdt_proj_fwd
    monkeypatch.setattr(mamba_inner, "xdt_proj_fwd", lambda *a, **k: (calls.append((a[0].dtype, k)), real(*a, **k))[1])
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o1 = m(h)
    assert calls and calls[0][0] == torch.bfloat16 and calls[0][1].get("conv") is not None
    monkeypatch.setattr(mamba_inner, "xdt_proj_fused_ok", lambda *a, **k: False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o2 = m(h)
    assert nerr(o1, o2) < 1e-2
