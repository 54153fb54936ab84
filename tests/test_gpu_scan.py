"""GPU parity: selective scan fwd/bwd through the C ABI vs the CPU oracle.

Tolerances (BASELINE.json north_star): 1e-3 for fp32 I/O, 1e-2 for bf16 I/O, measured as
max|got - want| / max(1, max|want|) per tensor (gradient sums over up to B*L*D terms are compared on
the same normalised scale).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import scan_ref
from oracle.gen_golden import scan_inputs

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


def run_hip(inp, device, dtype=torch.float32, grads=True):
    from si_mamba_amd import selective_scan_fn
    act = ("u", "delta", "z", "B", "C")
    t = {}
    for k, v in inp.items():
        if v is None or k == "dout":
            t[k] = v
            continue
        x = v.to(device)
        if k in act:
            x = x.to(dtype)
        t[k] = x.requires_grad_(grads)
    out, last = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"],
                                  delta_softplus=True, return_last_state=True)
    g = {}
    if grads:
        out.backward(inp["dout"].to(device).to(dtype))
        g = {k: t[k].grad for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias") if t.get(k) is not None}
    return out, last, g


def run_oracle(inp, dtype=torch.float32):
    """Oracle on the same (dtype-rounded) inputs, fp32 accumulation."""
    act = ("u", "delta", "z", "B", "C")
    t = {}
    for k, v in inp.items():
        if v is None or k == "dout":
            t[k] = v
            continue
        x = v.to(dtype).float() if k in act else v.clone()
        t[k] = x.requires_grad_(True)
    out, last = scan_ref.selective_scan_ref(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"],
                                            t["delta_bias"], delta_softplus=True, return_last_state=True)
    out.backward(inp["dout"].to(dtype).float())
    g = {k: t[k].grad for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias") if t.get(k) is not None}
    return out, last, g


@pytest.mark.parametrize("name", ["scan_cfg1", "scan_l128", "scan_multichunk_ragged", "scan_odd"])
def test_scan_golden(name, device):
    g = load_golden(name)
    inp = {k: (torch.from_numpy(g[k]) if k in g else None)
           for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "dout")}
    out, last, grads = run_hip(inp, device)
    assert nerr(out, torch.from_numpy(g["out"])) < 1e-3
    assert nerr(last, torch.from_numpy(g["last_state"])) < 1e-3
    for k, v in grads.items():
        assert nerr(v, torch.from_numpy(g["grad_" + k])) < 1e-3, k


SHAPES = [
    # batch, dim, L, N      what it exercises
    (2, 256, 64, 16),       # BASELINE config 1 scan shape (kItems = 4 path)
    (3, 100, 128, 16),      # dim not a multiple of 16 / of the workgroup tile
    (2, 32, 129, 16),       # one step into the second chunk
    (1, 48, 208, 16),       # MAE encoder length (26 visible x 4 x 2), ragged last chunk
    (2, 24, 512, 16),       # 4 chunks
    (1, 16, 1, 16),         # single step
    (2, 17, 50, 7),         # unaligned L (scalar access path), dstate < 16
    (1, 8, 3, 1),           # tiny everything
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_random_shapes(shape, dtype, device):
    b, d, L, N = shape
    inp = scan_inputs(b, d, L, N, seed=100 + L)
    out, last, grads = run_hip(inp, device, dtype)
    wout, wlast, wgrads = run_oracle(inp, dtype)
    tol = TOL[dtype]
    assert nerr(out, wout) < tol
    assert nerr(last, wlast) < tol
    for k in wgrads:
        assert nerr(grads[k], wgrads[k]) < tol, k


@pytest.mark.parametrize("with_z,with_D,with_bias", [(False, True, True), (True, False, False), (False, False, False)])
def test_scan_optional_operands(with_z, with_D, with_bias, device):
    inp = scan_inputs(2, 40, 96, 16, seed=7, with_z=with_z, with_D=with_D, with_bias=with_bias)
    out, _, grads = run_hip(inp, device)
    wout, _, wgrads = run_oracle(inp)
    assert nerr(out, wout) < 1e-3
    assert set(grads) == set(wgrads)
    for k in wgrads:
        assert nerr(grads[k], wgrads[k]) < 1e-3, k


def test_scan_no_softplus_and_grouped_bc(device):
    from si_mamba_amd import selective_scan_fn
    inp = scan_inputs(2, 16, 70, 16, seed=9)
    dl = torch.nn.functional.softplus(inp["delta"])         # positive step sizes, softplus off
    got = selective_scan_fn(inp["u"].to(device), dl.to(device), inp["A"].to(device),
                            inp["B"][:, None].to(device), inp["C"][:, None].to(device), None, None, None, False)
    want = scan_ref.selective_scan_ref(inp["u"], dl, inp["A"], inp["B"], inp["C"])
    assert nerr(got, want) < 1e-3


def test_scan_empty_inputs(device):
    from si_mamba_amd import selective_scan_fn
    z = torch.zeros(0, 16, 32, device=device)
    out = selective_scan_fn(z, z, -torch.ones(16, 16, device=device), torch.zeros(0, 16, 32, device=device),
                            torch.zeros(0, 16, 32, device=device))
    assert out.shape == (0, 16, 32)


# ---- BASELINE-size checks through size-independent properties -------------------------------------
def _full(device, b, d, L, seed):
    inp = scan_inputs(b, d, L, 16, seed=seed)
    return {k: (v.to(device) if v is not None else None) for k, v in inp.items()}


@pytest.mark.parametrize("shape", [(256, 768, 128), (64, 768, 1024)])
def test_scan_full_size_properties(shape, device):
    """Headline micro-shape and the model-level shape: (i) linear in u, (ii) a random subset of rows
    equals the oracle run on just those rows, (iii) batch permutation equivariance."""
    from si_mamba_amd import selective_scan_fn
    b, d, L = shape
    t = _full(device, b, d, L, seed=1)
    f = lambda u, bi=slice(None): selective_scan_fn(u, t["delta"][bi], t["A"], t["B"][bi], t["C"][bi], t["D"],
                                                    t["z"][bi], t["delta_bias"], True)
    out = f(t["u"])
    assert torch.isfinite(out).all()
    u2 = torch.randn_like(t["u"])
    lin = f(t["u"] + 2.0 * u2) - (out + 2.0 * f(u2))
    assert lin.abs().max().item() < 2e-3 * max(1.0, out.abs().max().item())
    gsel = torch.Generator().manual_seed(0)
    bs = torch.randint(0, b, (3,), generator=gsel).tolist()
    ds = torch.randint(0, d, (24,), generator=gsel)
    for bi in bs:
        want = scan_ref.selective_scan_ref(t["u"][bi:bi + 1, ds].cpu(), t["delta"][bi:bi + 1, ds].cpu(),
                                           t["A"][ds].cpu(), t["B"][bi:bi + 1].cpu(), t["C"][bi:bi + 1].cpu(),
                                           t["D"][ds].cpu(), t["z"][bi:bi + 1, ds].cpu(),
                                           t["delta_bias"][ds].cpu(), True)
        assert nerr(out[bi:bi + 1, ds], want) < 1e-3
    perm = torch.randperm(b, generator=gsel).to(device)
    out_p = selective_scan_fn(t["u"][perm], t["delta"][perm], t["A"], t["B"][perm], t["C"][perm], t["D"],
                              t["z"][perm], t["delta_bias"], True)
    assert torch.equal(out_p, out[perm])


def test_scan_backward_full_size_against_row_subset(device):
    """Backward at the headline shape: du/ddelta/dz rows vs the oracle on a row subset (those gradients are
    per-row), and dA/dD/dbias/dB/dC as sums checked on a reduced-batch replay of the same rows."""
    from si_mamba_amd import selective_scan_fn
    b, d, L = 64, 768, 128
    t = _full(device, b, d, L, seed=2)
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("u", "delta", "z")}
    out = selective_scan_fn(leaves["u"], leaves["delta"], t["A"], t["B"], t["C"], t["D"], leaves["z"],
                            t["delta_bias"], True)
    out.backward(t["dout"])
    bi, ds = 5, torch.arange(0, d, 37)
    sub = {k: t[k][bi:bi + 1, ds].cpu().clone().requires_grad_(True) for k in ("u", "delta", "z")}
    want = scan_ref.selective_scan_ref(sub["u"], sub["delta"], t["A"][ds].cpu(), t["B"][bi:bi + 1].cpu(),
                                       t["C"][bi:bi + 1].cpu(), t["D"][ds].cpu(), sub["z"],
                                       t["delta_bias"][ds].cpu(), True)
    want.backward(t["dout"][bi:bi + 1, ds].cpu())
    for k in ("u", "delta", "z"):
        assert nerr(leaves[k].grad[bi:bi + 1, ds], sub[k].grad) < 1e-3, k


# ---- the one-lane-per-channel forward (taken when batch*dim >= 98304 and rows are pack-aligned) ----------
@pytest.mark.parametrize("L,dtype,split", [(16, torch.float32, 2), (40, torch.float32, 1), (132, torch.float32, 2),
                                           (260, torch.float32, 4), (260, torch.float32, 1),
                                           (128, torch.float32, 0), (136, torch.bfloat16, 2),
                                           (264, torch.bfloat16, 4)])
def test_scan_seq_kernel_path(L, dtype, split, device, monkeypatch):
    """Same parity bar as the row-scan kernel, on a row subset (the full tensor is too slow for the CPU
    oracle), including the chunk checkpoints it hands to the backward (L > 128) and the final state."""
    from si_mamba_amd import _lib, selective_scan_fn
    monkeypatch.setenv("SIMAMBA_SEQ_FWD", "1")
    if split:      # a channel's 16 states are split over `split` adjacent lanes; 0 = the dispatcher's own choice
        monkeypatch.setenv("SIMAMBA_SEQ_LPC", str(split))
    B, D, N = 128, 768, 16          # batch * dim = 98304 rows: the dispatcher's threshold for this kernel
    inp = scan_inputs(B, D, L, N, seed=L)
    t = {k: v.to(device) for k, v in inp.items()}
    for k in ("u", "delta", "z", "B", "C", "dout"):
        t[k] = t[k].to(dtype)
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("u", "delta", "z")}
    out, last = selective_scan_fn(leaves["u"], leaves["delta"], t["A"], t["B"], t["C"], t["D"], leaves["z"],
                                  t["delta_bias"], True, True)
    out.backward(t["dout"])
    tol = TOL[dtype]
    g = torch.Generator().manual_seed(1)
    ds = torch.randperm(D, generator=g)[:40]
    for bi in (0, 77, 127):
        sub = {k: t[k][bi:bi + 1, ds].float().cpu().clone().requires_grad_(True) for k in ("u", "delta", "z")}
        want, wlast = scan_ref.selective_scan_ref(sub["u"], sub["delta"], t["A"][ds].cpu(),
                                                  t["B"][bi:bi + 1].float().cpu(), t["C"][bi:bi + 1].float().cpu(),
                                                  t["D"][ds].cpu(), sub["z"], t["delta_bias"][ds].cpu(), True, True)
        want.backward(t["dout"][bi:bi + 1, ds].float().cpu())
        assert nerr(out[bi:bi + 1, ds], want) < tol
        assert nerr(last[bi:bi + 1, ds], wlast) < tol
        for k in ("u", "delta", "z"):
            assert nerr(leaves[k].grad[bi:bi + 1, ds], sub[k].grad) < tol, k


def test_scan_strided_operands(device):
    """z as a batch-strided half of a (B, 2D, L) tensor and B/C as token-major views of a (B, L, R+2N)
    tensor -- the layouts the mixer produces -- are read in place and match the contiguous call."""
    from si_mamba_amd import selective_scan_fn
    inp = scan_inputs(3, 48, 160, 16, seed=21)
    t = {k: v.to(device) for k, v in inp.items()}
    xz = torch.randn(3, 96, 160, device=device)
    xz[:, 48:] = t["z"]
    x_dbl = torch.randn(3, 160, 8 + 32, device=device)
    x_dbl[:, :, 8:24] = t["B"].transpose(1, 2)
    x_dbl[:, :, 24:] = t["C"].transpose(1, 2)
    zs = xz[:, 48:].requires_grad_(False)
    Bs, Cs = x_dbl[:, :, 8:24].transpose(1, 2), x_dbl[:, :, 24:].transpose(1, 2)
    assert not zs.is_contiguous() and not Bs.is_contiguous()
    u = t["u"].clone().requires_grad_(True)
    u2 = t["u"].clone().requires_grad_(True)
    a = selective_scan_fn(u, t["delta"], t["A"], Bs, Cs, t["D"], zs, t["delta_bias"], True)
    b = selective_scan_fn(u2, t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)
    assert torch.equal(a, b)
    a.backward(t["dout"]); b.backward(t["dout"])
    torch.testing.assert_close(u.grad, u2.grad, rtol=1e-5, atol=1e-5)
