"""GPU parity of the scan kernels that form delta themselves (simamba_selective_scan_dt_fwd / _bwd, SURVEY 2a "best fused
into the scan prologue") against the path that materialises it (simamba_xdt_proj_fwd + simamba_selective_scan_fwd / _bwd)
and against the oracle.

The in-kernel delta uses the matrix-pipe instruction and k order of the xdt kernel, so the forward output is required
to be BIT-IDENTICAL to the materialised path; the backward's per-element gradients (du, ddelta, dz: no atomics) likewise,
its accumulators (float atomics in both paths) to 1e-5.
"""
import pytest
import torch

from oracle import scan_ref

pytestmark = pytest.mark.gpu


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


def _operands(B, D, L, R, dtype, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    N, S = 16, R + 32
    u = torch.randn(B, D, L, generator=g)
    z = torch.randn(B, D, L, generator=g)
    xdbl = torch.randn(B, L, S, generator=g)
    xdbl[:, :, :R] *= 0.5
    wdt = torch.randn(D, R, generator=g) * R ** -0.5
    A = -torch.exp(torch.log(torch.arange(1, N + 1).float())[None].expand(D, -1) + 0.1 * torch.randn(D, N, generator=g))
    Dp = torch.ones(D) + 0.1 * torch.randn(D, generator=g)
    bias = torch.log(torch.expm1(torch.exp(torch.rand(D, generator=g) * 4.6 - 6.9)))
    dout = torch.randn(B, D, L, generator=g)
    t = dict(u=u.to(dtype), z=z.to(dtype), xdbl=xdbl.to(dtype), wdt=wdt.to(dtype), dout=dout.to(dtype))
    t = {k: v.to(device).contiguous() for k, v in t.items()}
    t.update(A=A.contiguous().to(device), D=Dp.to(device), bias=bias.to(device))
    return t


def _run(t, fused, variant=0):
    from si_mamba_amd import _lib
    lib = _lib.load()
    dev = t["u"].device
    B, D, L = t["u"].shape
    R = t["wdt"].shape[1]
    N = 16
    code = _lib.dtype_code(t["u"].dtype)
    st = _lib.stream_ptr(dev)
    x = t["xdbl"]
    Bv, Cv = x[:, :, R:R + N], x[:, :, R + N:]
    nck = lib.simamba_scan_ckpt_floats(B, D, L, N, _lib.CKPT_SEQ)
    ck = torch.empty(nck, device=dev) if nck else None
    out = torch.empty_like(t["u"])
    du, dd, dz = (torch.empty_like(t["u"]) for _ in range(3))
    acc = _lib.scan_bwd_accumulators(B, D, L, N, True, True, dev)
    if fused:
        rc = lib.simamba_selective_scan_dt_fwd(t["u"].data_ptr(), x.data_ptr(), t["wdt"].data_ptr(), t["A"].data_ptr(),
                                               t["D"].data_ptr(), t["z"].data_ptr(), t["bias"].data_ptr(), out.data_ptr(),
                                               _lib.ptr(ck), None, B, D, L, N, R, code, 0, 0, 0, _lib.CKPT_SEQ, variant, st)
        assert rc == 0, rc
        rc = lib.simamba_selective_scan_dt_bwd(t["u"].data_ptr(), x.data_ptr(), t["wdt"].data_ptr(), t["A"].data_ptr(),
                                               t["D"].data_ptr(), t["z"].data_ptr(), t["bias"].data_ptr(),
                                               t["dout"].data_ptr(), _lib.ptr(ck), du.data_ptr(), dd.data_ptr(),
                                               acc[0].data_ptr(), acc[1].data_ptr(), acc[2].data_ptr(), acc[3].data_ptr(),
                                               dz.data_ptr(), acc[4].data_ptr(), B, D, L, N, R, code, 0, 0, 0, 0, st)
        assert rc == 0, rc
        delta = None
    else:
        wx = torch.zeros(R + 2 * N, D, device=dev, dtype=t["u"].dtype)      # only the delta product of the xdt kernel is used
        delta = torch.empty_like(t["u"])
        # delta = wdt @ dt through the xdt kernel's own phase 2: feed it an x whose x_proj IS the given x_dbl -- not
        # possible in general, so the reference delta is formed by the library GEMM on the same operands instead and
        # compared separately; the scan below reads the tensor the fused kernels must reproduce
        delta = torch.bmm(t["wdt"].unsqueeze(0).expand(B, -1, -1), x[:, :, :R].transpose(1, 2)).contiguous()
        rc = lib.simamba_selective_scan_fwd(t["u"].data_ptr(), delta.data_ptr(), t["A"].data_ptr(), Bv.data_ptr(),
                                            Cv.data_ptr(), t["D"].data_ptr(), t["z"].data_ptr(), t["bias"].data_ptr(),
                                            out.data_ptr(), _lib.ptr(ck), None, B, D, L, N, code, 1, 0, x.stride(0), 1,
                                            x.stride(1), _lib.CKPT_SEQ, variant, st)
        assert rc == 0, rc
        rc = lib.simamba_selective_scan_bwd(t["u"].data_ptr(), delta.data_ptr(), t["A"].data_ptr(), Bv.data_ptr(),
                                            Cv.data_ptr(), t["D"].data_ptr(), t["z"].data_ptr(), t["bias"].data_ptr(),
                                            t["dout"].data_ptr(), _lib.ptr(ck), du.data_ptr(), dd.data_ptr(),
                                            acc[0].data_ptr(), acc[1].data_ptr(), acc[2].data_ptr(), acc[3].data_ptr(),
                                            dz.data_ptr(), acc[4].data_ptr(), B, D, L, N, code, 1, 0, 0, x.stride(0), 1,
                                            x.stride(1), _lib.CKPT_SEQ, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    return dict(out=out, du=du, ddelta=dd, dz=dz, dA=acc[0].clone(), dB=acc[1].clone(), dC=acc[2].clone(),
                dD=acc[3].clone(), dbias=acc[4].clone(), delta=delta)


@pytest.mark.parametrize("shape", [(2, 64, 64, 8), (1, 128, 208, 16), (2, 192, 96, 24), (3, 64, 40, 24), (1, 64, 16, 24)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("variant", [2, 4])
def test_dt_scan_matches_oracle_and_materialised_path(shape, dtype, variant, device):
    B, D, L, R = shape
    if dtype == torch.bfloat16 and (L % 8 or R % 8):
        pytest.skip("bf16 packs are 8 elements")
    t = _operands(B, D, L, R, dtype, device, seed=L + R)
    got = _run(t, True, variant)
    # ---- oracle on the same (dtype-rounded) operands; delta rounded once where the materialised tensor would be
    f = {k: v.float().cpu() for k, v in t.items()}
    dl = torch.einsum("dr,blr->bdl", f["wdt"], f["xdbl"][:, :, :R])
    if dtype == torch.bfloat16:
        dl = dl.bfloat16().float()
    leaf = {k: f[k].clone().requires_grad_(True) for k in ("u", "z", "A", "D", "bias")}
    dl = dl.requires_grad_(True)
    Bm = f["xdbl"][:, :, R:R + 16].transpose(1, 2).clone().requires_grad_(True)
    Cm = f["xdbl"][:, :, R + 16:].transpose(1, 2).clone().requires_grad_(True)
    want = scan_ref.selective_scan_ref(leaf["u"], dl, leaf["A"], Bm, Cm, leaf["D"], leaf["z"], leaf["bias"],
                                       delta_softplus=True)
    want.backward(f["dout"])
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    assert nerr(got["out"], want) < tol
    for k, w in (("du", leaf["u"].grad), ("ddelta", dl.grad), ("dz", leaf["z"].grad), ("dA", leaf["A"].grad),
                 ("dB", Bm.grad), ("dC", Cm.grad), ("dD", leaf["D"].grad), ("dbias", leaf["bias"].grad)):
        assert nerr(got[k], w) < tol, k
    # ---- the materialised path through the same scan kernels: fp32 delta from a library GEMM differs from the MFMA
    # chain in the last bits, so this comparison is close, not exact (the exact one is the mixer-level test below)
    ref = _run(t, False, variant)
    for k in ("out", "du", "ddelta", "dz", "dA", "dB", "dC", "dD", "dbias"):
        assert nerr(got[k], ref[k]) < (1e-5 if dtype == torch.float32 else 1e-2), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d_model,B,L", [(384, 2, 64), (128, 3, 96), (256, 1, 208)])
def test_mixer_with_in_kernel_delta_is_bit_identical_to_materialised_delta(d_model, B, L, dtype, device):
    """The whole mixer with delta formed in the scan kernels vs. the same mixer with the xdt kernel storing delta (which
    is what round 2 shipped): the forward output bit for bit, every gradient to 1e-5 of its scale (the per-element
    gradients of the scan are bit-identical too; the weight gradients go through float atomics in both paths)."""
    from si_mamba_amd import Mamba, _lib
    torch.manual_seed(0)
    m = Mamba(d_model, layer_idx=0).to(device)
    h = torch.randn(B, L, d_model, device=device)
    w = torch.randn(B, L, d_model, device=device)
    res = {}
    for fused in (True, False):
        before = _lib.counters.get("scan_dt_fwd", 0)
        m.zero_grad(set_to_none=True)
        x = h.clone().requires_grad_(True)
        with _lib.scan_ckpt(_lib.CKPT_SEQ), _lib.scan_fuse_dt(fused), torch.autocast("cuda", dtype=torch.bfloat16,
                                                                                    enabled=dtype == torch.bfloat16):
            out = m(x)
        (out.float() * w).sum().backward()
        assert (_lib.counters.get("scan_dt_fwd", 0) - before) == (1 if fused else 0)       # the route under test ran
        res[fused] = (out.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])
    scale = 1e-5 if dtype == torch.float32 else 2e-2        # bf16: dx passes through atomically accumulated dB/dC
    assert nerr(res[True][1], res[False][1]) < scale
    for k in res[True][2]:
        assert nerr(res[True][2][k], res[False][2][k]) < scale, k
