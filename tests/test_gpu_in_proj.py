"""GPU parity of the hand-written bf16 in_proj kernel (csrc/in_proj_bf16.hip; reference: the mixer's
self.in_proj, reached from models/block.py:72) against float64 products of the bf16 operands and against the library
GEMM route it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


def _call(x, w, device, dtype=torch.bfloat16):
    from si_mamba_amd import _lib
    lib = _lib.load()
    B, L, C = x.shape
    M = w.shape[0]
    xz = torch.full((B, M, L), float("nan"), device=device, dtype=dtype)
    rc = lib.simamba_in_proj_fwd(x.data_ptr(), w.data_ptr(), xz.data_ptr(), B, L, C, M, _lib.dtype_code(x.dtype),
                                 _lib.stream_ptr(x.device))
    return rc, xz


@pytest.mark.parametrize("B,L,C,M", [(2, 256, 384, 1536), (1, 1024, 384, 1536), (3, 72, 128, 512), (2, 520, 64, 64),
                                     (1, 8, 192, 96), (2, 264, 256, 1024), (1, 512, 320, 640)])
def test_in_proj_kernel_matches_float64_product(B, L, C, M, device):
    """Through the C ABI; ragged tiles (L % 256 != 0), every K depth the kernel instantiates, one rounding to bf16."""
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, L, C, generator=g).bfloat16().to(device)
    w = (torch.randn(M, C, generator=g) * C ** -0.5).bfloat16().to(device)
    rc, xz = _call(x, w, device)
    assert rc == 0
    want = torch.einsum("jc,blc->bjl", w.double().cpu(), x.double().cpu())
    assert torch.isfinite(xz.float()).all()
    # fp32 accumulation + one bf16 rounding: half a bf16 ulp of the value, plus accumulation noise
    err = (xz.double().cpu() - want).abs()
    assert (err <= 2.0 ** -8 * want.abs() + 1e-3).all()


@pytest.mark.parametrize("B,L,C,M", [(2, 128, 384, 1536), (1, 1024, 384, 768), (3, 72, 128, 512), (2, 260, 64, 64),
                                     (1, 4, 192, 96), (2, 132, 256, 1024), (1, 512, 320, 640)])
def test_in_proj_f32_kernel_matches_float64_product(B, L, C, M, device):
    """The fp32 form (csrc/in_proj_f32.hip, exact-fp32 MFMA): every K depth it instantiates, ragged 128-token tiles;
    fp32 accumulation error only (north-star 1e-3; observed ~1e-6)."""
    g = torch.Generator().manual_seed(B * 1000 + L + 7)
    x = torch.randn(B, L, C, generator=g).to(device)
    w = (torch.randn(M, C, generator=g) * C ** -0.5).to(device)
    rc, xz = _call(x, w, device, torch.float32)
    assert rc == 0
    want = torch.einsum("jc,blc->bjl", w.double().cpu(), x.double().cpu())
    assert torch.isfinite(xz).all()
    assert nerr(xz, want) < 2e-6 * max(1.0, C ** 0.5 / 8)


def test_in_proj_kernel_refuses_what_it_does_not_take(device):
    from si_mamba_amd import _lib
    x = torch.zeros(1, 16, 100, device=device, dtype=torch.bfloat16)
    w = torch.zeros(64, 100, device=device, dtype=torch.bfloat16)
    rc, _ = _call(x, w, device)                                     # C % 64
    assert rc == _lib.load().simamba_in_proj_fwd(None, None, None, 1, 16, 100, 64, _lib.BF16, None) != 0
    x = torch.zeros(1, 12, 64, device=device, dtype=torch.bfloat16)   # L % 8
    w = torch.zeros(64, 64, device=device, dtype=torch.bfloat16)
    assert _call(x, w, device)[0] != 0
    x = torch.zeros(0, 16, 64, device=device, dtype=torch.bfloat16)   # empty batch: nothing to do
    assert _call(x, w, device)[0] == 0


@pytest.mark.parametrize("B,L", [(4, 256), (2, 1024)])
def test_in_proj_fn_hand_route_matches_library_route_fp32(B, L, device):
    """fp32 (no autocast): hand kernel against the library GEMM at fp32 accumulation noise, backward identical."""
    from si_mamba_amd import _lib
    from si_mamba_amd.mamba_inner import in_proj_fn
    g = torch.Generator().manual_seed(6)
    h = torch.randn(B, L, 384, generator=g).to(device)
    w = (torch.randn(1536, 384, generator=g) * 384 ** -0.5).to(device)
    dout = torch.randn(B, 1536, L, generator=g).to(device)
    res = {}
    for hand in (True, False):
        hh, ww = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
        _lib.counters.pop("in_proj_hand", None)
        with _lib.hand_in_proj(hand):
            xz = in_proj_fn(hh, ww)
        assert _lib.counters.get("in_proj_hand", 0) == (1 if hand else 0)
        assert xz.dtype == torch.float32 and xz.shape == (B, 1536, L)
        (xz * dout).sum().backward()
        res[hand] = (xz.detach(), hh.grad, ww.grad)
    assert nerr(res[True][0], res[False][0]) < 1e-5
    assert nerr(res[True][1], res[False][1]) < 1e-6 and nerr(res[True][2], res[False][2]) < 1e-6


@pytest.mark.parametrize("B,L", [(4, 256), (2, 1024)])
def test_in_proj_fn_hand_route_matches_library_route(B, L, device):
    """in_proj_fn under bf16 autocast: the hand kernel against the library GEMM (forward within bf16 rounding flips;
    the backward is the same library code on both routes), and the counter shows which route ran."""
    from si_mamba_amd import _lib
    from si_mamba_amd.mamba_inner import in_proj_fn
    g = torch.Generator().manual_seed(5)
    h = torch.randn(B, L, 384, generator=g).to(device)
    w = (torch.randn(1536, 384, generator=g) * 384 ** -0.5).to(device)
    dout = torch.randn(B, 1536, L, generator=g).to(device)
    res = {}
    for hand in (True, False):
        hh, ww = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
        _lib.counters.pop("in_proj_hand", None)
        with _lib.hand_in_proj(hand), torch.autocast("cuda", dtype=torch.bfloat16):
            xz = in_proj_fn(hh, ww)
        assert _lib.counters.get("in_proj_hand", 0) == (1 if hand else 0)
        assert xz.dtype == torch.bfloat16 and xz.shape == (B, 1536, L)
        (xz.float() * dout).sum().backward()
        res[hand] = (xz.detach(), hh.grad, ww.grad)
    assert nerr(res[True][0], res[False][0]) < 1e-2
    assert ((res[True][0].float() - res[False][0].float()).abs() > 1e-6).float().mean() < 5e-2
    assert nerr(res[True][1], res[False][1]) < 1e-6 and nerr(res[True][2], res[False][2]) < 1e-6
