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
_full_cache = {}


def _full(device, b, d, L, seed):
    """Full-size synthetic inputs on the device (generated once per shape and seed: the seeded CPU generator takes
    tens of seconds at these sizes)."""
    key = (b, d, L, seed)
    if key not in _full_cache:
        _full_cache.clear()                     # one shape resident at a time (1.2 GB at the model shape)
        inp = scan_inputs(b, d, L, 16, seed=seed)
        _full_cache[key] = {k: (v.to(device) if v is not None else None) for k, v in inp.items()}
    return dict(_full_cache[key])


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


def _oracle_grads(t, bsel, dsel, dtype, threads=None):
    """Oracle forward + backward restricted to samples ``bsel`` and channels ``dsel`` (index tensors / slices) of the
    device tensors ``t``; activations rounded to ``dtype`` like the kernel's inputs, fp32 accumulation.
    ``threads``: CPU threads for the step loop (its per-step tensors are tiny for a channel subset: one thread is
    ~20x faster there than a thread pool that synchronises 10 times per step)."""
    if threads is not None:
        keep = torch.get_num_threads()
        torch.set_num_threads(threads)
        try:
            return _oracle_grads(t, bsel, dsel, dtype)
        finally:
            torch.set_num_threads(keep)
    act = lambda x: x.to(dtype).float().cpu()
    leaf = {
        "u": act(t["u"][bsel][:, dsel]), "delta": act(t["delta"][bsel][:, dsel]), "z": act(t["z"][bsel][:, dsel]),
        "B": act(t["B"][bsel]), "C": act(t["C"][bsel]),
        "A": t["A"][dsel].cpu().clone(), "D": t["D"][dsel].cpu().clone(), "delta_bias": t["delta_bias"][dsel].cpu().clone(),
    }
    for v in leaf.values():
        v.requires_grad_(True)
    out = scan_ref.selective_scan_ref(leaf["u"], leaf["delta"], leaf["A"], leaf["B"], leaf["C"], leaf["D"], leaf["z"],
                                      leaf["delta_bias"], True)
    out.backward(act(t["dout"][bsel][:, dsel]))
    return {k: v.grad for k, v in leaf.items()}


@pytest.mark.parametrize("shape,dtype", [((64, 768, 1024), torch.float32), ((64, 768, 1024), torch.bfloat16),
                                          ((128, 768, 256), torch.float32)])
def test_scan_backward_full_size_reductions(shape, dtype, device):
    """Backward at the model shape (64,768,1024,16) -- what every Mamba block of the bench step runs: several channel
    passes per workgroup, 8 chunks with carried adjoint states, the cross-wave LDS sum and the float-atomic dB / dC
    flush from >= 512 workgroups with the XCD relabelling -- and at two shapes with other pass / chunk counts.
      * du / ddelta / dz (per element): rows of a few samples x channels against the oracle on exactly those rows;
      * dA / dD / ddelta_bias (sums over batch and time): a channel subset against the oracle run on ALL samples of
        those channels;
      * dB / dC (sums over channels): two samples against the oracle run on ALL 768 channels of those samples."""
    from si_mamba_amd import selective_scan_fn
    b, d, L = shape
    t = _full(device, b, d, L, seed=2)
    act = ("u", "delta", "z", "B", "C", "dout")
    tt = {k: (v.to(dtype) if k in act else v) for k, v in t.items()}
    leaves = {k: tt[k].clone().requires_grad_(True) for k in ("u", "delta", "z", "B", "C", "A", "D", "delta_bias")}
    out = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                            leaves["z"], leaves["delta_bias"], True)
    out.backward(tt["dout"])
    tol = TOL[dtype]
    got = {k: v.grad for k, v in leaves.items()}
    # (i) parameter gradients of a channel subset, summed over every sample and step
    ds = torch.tensor([0, 5, 17, 255, 256, 400, 766, 767], device=device)
    want = _oracle_grads(tt, slice(None), ds, dtype, threads=1)
    for k, name in (("A", "dA"), ("D", "dD"), ("delta_bias", "ddelta_bias")):
        assert nerr(got[k][ds], want[k]) < tol, name
    for k in ("u", "delta", "z"):
        assert nerr(got[k][:, ds], want[k]) < tol, k
    # (ii) dB / dC of two samples, summed over all channels
    bs = torch.tensor([3, b - 1], device=device)
    want = _oracle_grads(tt, bs, slice(None), dtype)
    for k in ("B", "C"):
        assert nerr(got[k][bs], want[k]) < tol, "d" + k
    for k in ("u", "delta", "z"):
        assert nerr(got[k][bs], want[k]) < tol, k
    # the remaining samples' dB / dC: finite, and no sample was skipped or flushed twice -- a sample's dC norm is
    # of the order of the checked ones
    nrm = got["C"].float().flatten(1).norm(dim=1)
    assert torch.isfinite(nrm).all() and (nrm > 0.25 * nrm[bs].min()).all() and (nrm < 4 * nrm[bs].max()).all()


# ---- the lanes-per-channel forward (the library's choice from 49 152 rows on; forced here through the ABI's variant) --
@pytest.mark.parametrize("L,dtype,split", [(16, torch.float32, 2), (40, torch.float32, 4), (132, torch.float32, 2),
                                           (260, torch.float32, 4), (260, torch.float32, 2), (32, torch.float32, 2),
                                           (128, torch.float32, 0), (136, torch.bfloat16, 2),
                                           (264, torch.bfloat16, 4), (1024, torch.float32, 4), (1024, torch.float32, 2)])
def test_scan_seq_kernel_path(L, dtype, split, device):
    """Same parity bar as the row-scan kernel, on a row subset (the full tensor is too slow for the CPU
    oracle), including the chunk checkpoints it hands to the backward (L > 128) and the final state.
    ``split``: lanes per channel (a channel's 16 states split over 2 or 4 adjacent lanes); 0 = the library's choice."""
    from si_mamba_amd import _lib, selective_scan_fn
    B, D, N = (128, 768, 16) if L < 1024 else (48, 768, 16)
    inp = scan_inputs(B, D, L, N, seed=L)
    t = {k: v.to(device) for k, v in inp.items()}
    for k in ("u", "delta", "z", "B", "C", "dout"):
        t[k] = t[k].to(dtype)
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("u", "delta", "z")}
    with _lib.scan_variant({0: _lib.SCAN_AUTO, 2: _lib.SCAN_LPC2, 4: _lib.SCAN_LPC4}[split]):
        out, last = selective_scan_fn(leaves["u"], leaves["delta"], t["A"], t["B"], t["C"], t["D"], leaves["z"],
                                      t["delta_bias"], True, True)
        # every variant computes the same function: the row-scan kernel agrees to rounding on the whole tensor
        with torch.no_grad(), _lib.scan_variant(_lib.SCAN_ROWSCAN):
            out_r, last_r = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"],
                                              t["delta_bias"], True, True)
    tol = TOL[dtype]
    assert nerr(out, out_r) < tol and nerr(last, last_r) < tol
    out.backward(t["dout"])
    g = torch.Generator().manual_seed(1)
    ds = torch.randperm(D, generator=g)[:40]
    for bi in (0, B // 2 + 13, B - 1):
        sub = {k: t[k][bi:bi + 1, ds].float().cpu().clone().requires_grad_(True) for k in ("u", "delta", "z")}
        want, wlast = scan_ref.selective_scan_ref(sub["u"], sub["delta"], t["A"][ds].cpu(),
                                                  t["B"][bi:bi + 1].float().cpu(), t["C"][bi:bi + 1].float().cpu(),
                                                  t["D"][ds].cpu(), sub["z"], t["delta_bias"][ds].cpu(), True, True)
        want.backward(t["dout"][bi:bi + 1, ds].float().cpu())
        assert nerr(out[bi:bi + 1, ds], want) < tol
        assert nerr(last[bi:bi + 1, ds], wlast) < tol
        for k in ("u", "delta", "z"):
            assert nerr(leaves[k].grad[bi:bi + 1, ds], sub[k].grad) < tol, k


@pytest.mark.parametrize("B,D,L,dtype", [(64, 768, 160, torch.float32), (64, 768, 1024, torch.float32),
                                         (32, 1280, 96, torch.float32), (12, 1024, 200, torch.float32),
                                         (64, 768, 136, torch.bfloat16)])
def test_scan_mixed_launch(B, D, L, dtype, device):
    """SIMAMBA_SCAN_MIX: one launch in which the first channels of every sample run two lanes per channel and the last
    ones four (an explicit variant: faster than two lanes per channel on its own at batch * dim = 49 152, slower
    where the mixer calls it, so the library's own choice stays two lanes -- csrc/scan_fwd_seq.hip).  Whole-tensor agreement with the
    row-scan kernel -- outputs, final state, and the gradients the backward forms from the chunk checkpoints this
    forward wrote -- plus the oracle on a few rows on either side of the seam.  (12, 1024): batch not a multiple of
    the 8 XCDs and every channel on four lanes; (32, 1280): seam at channel 1024."""
    from si_mamba_amd import _lib, selective_scan_fn
    inp = scan_inputs(B, D, L, 16, seed=B + L)
    t = {k: v.to(device) for k, v in inp.items()}
    for k in ("u", "delta", "z", "B", "C", "dout"):
        t[k] = t[k].to(dtype)
    res = {}
    for name, variant in (("mix", _lib.SCAN_MIX), ("row", _lib.SCAN_ROWSCAN)):
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("u", "delta", "z")}
        with _lib.scan_variant(variant):
            out, last = selective_scan_fn(leaves["u"], leaves["delta"], t["A"], t["B"], t["C"], t["D"], leaves["z"],
                                          t["delta_bias"], True, True)
        out.backward(t["dout"])
        res[name] = (out.detach(), last.detach(), {k: v.grad for k, v in leaves.items()})
    tol = TOL[dtype]
    assert nerr(res["mix"][0], res["row"][0]) < tol and nerr(res["mix"][1], res["row"][1]) < tol
    for k in ("u", "delta", "z"):
        assert nerr(res["mix"][2][k], res["row"][2][k]) < tol, k
    if L <= 200:
        ds = torch.tensor([0, 31, 32, D // 2, D - 257, D - 256, D - 255, D - 17, D - 16, D - 1])
        for bi in (0, B - 1):
            want, wlast = scan_ref.selective_scan_ref(
                t["u"][bi:bi + 1, ds].float().cpu(), t["delta"][bi:bi + 1, ds].float().cpu(), t["A"][ds].cpu(),
                t["B"][bi:bi + 1].float().cpu(), t["C"][bi:bi + 1].float().cpu(), t["D"][ds].cpu(),
                t["z"][bi:bi + 1, ds].float().cpu(), t["delta_bias"][ds].cpu(), True, True)
            assert nerr(res["mix"][0][bi:bi + 1, ds], want) < tol
            assert nerr(res["mix"][1][bi:bi + 1, ds], wlast) < tol


def test_scan_mixed_launch_refused_where_it_does_not_apply(device):
    """An explicit MIX request on a shape with no half round of waves to fill is refused by name."""
    from si_mamba_amd import _lib, selective_scan_fn
    t = {k: v.to(device) for k, v in scan_inputs(48, 768, 64, 16, seed=5).items()}   # 1152 waves: 128 over, 85.3 channels a sample
    with torch.no_grad(), _lib.scan_variant(_lib.SCAN_MIX), pytest.raises(RuntimeError, match="variant"):
        selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)


@pytest.mark.parametrize("variant", [2, 4])
@pytest.mark.parametrize("layout", ["token_major", "odd_strides", "ragged_dim"])
def test_scan_seq_kernel_operand_layouts(variant, layout, device):
    """The lanes-per-channel kernel's two B / C staging routes (16-byte packs along time and along the state), strides
    that keep no pack aligned, a z read through a batch stride, and a channel count that is not a multiple of a
    wave's channels: bit-for-bit what the same kernel gives on contiguous copies."""
    from si_mamba_amd import _lib, selective_scan_fn
    b, d, L, N = 3, (100 if layout == "ragged_dim" else 96), 160, 16
    inp = scan_inputs(b, d, L, N, seed=77)
    t = {k: v.to(device) for k, v in inp.items()}
    if layout == "token_major":            # the mixer's layout: B | C inside the (B, L, R + 2N) x_proj output
        x_dbl = torch.randn(b, L, 24 + 32, device=device)
        x_dbl[:, :, 24:40] = t["B"].transpose(1, 2)
        x_dbl[:, :, 40:] = t["C"].transpose(1, 2)
        Bs, Cs = x_dbl[:, :, 24:40].transpose(1, 2), x_dbl[:, :, 40:].transpose(1, 2)
    elif layout == "odd_strides":          # time stride 1 but a state stride of L + 1: no 16-byte pack stays aligned
        pad = torch.randn(2, b, N, L + 1, device=device)
        pad[0, :, :, :L], pad[1, :, :, :L] = t["B"], t["C"]
        Bs, Cs = pad[0, :, :, :L], pad[1, :, :, :L]
    else:
        Bs, Cs = t["B"], t["C"]
    xz = torch.randn(b, 2 * d, L, device=device)
    xz[:, d:] = t["z"]
    zs = xz[:, d:]
    assert not zs.is_contiguous()
    if layout == "odd_strides":
        # no aligned pack of B / C exists: an explicit request is refused by name, the library's own choice
        # (variant AUTO) serves the operands through the row-scan kernel's element-wise gather
        with torch.no_grad(), _lib.scan_variant(variant), pytest.raises(RuntimeError, match="variant"):
            selective_scan_fn(t["u"], t["delta"], t["A"], Bs, Cs, t["D"], zs, t["delta_bias"], True, True)
        variant = _lib.SCAN_AUTO
    with torch.no_grad(), _lib.scan_variant(variant):
        got, glast = selective_scan_fn(t["u"], t["delta"], t["A"], Bs, Cs, t["D"], zs, t["delta_bias"], True, True)
        want, wlast = selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"],
                                        True, True)
    assert torch.equal(got, want) and torch.equal(glast, wlast)
    ref = scan_ref.selective_scan_ref(inp["u"][:1], inp["delta"][:1], inp["A"], inp["B"][:1], inp["C"][:1], inp["D"],
                                      inp["z"][:1], inp["delta_bias"], True)
    assert nerr(got[:1], ref) < 1e-3


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


def test_scan_backward_writes_nothing_outside_its_accumulators(device):
    """The five fp32 accumulators handed over as SEPARATE allocations with sentinel tensors between them (the
    caching allocator packs small blocks back to back): the library's zero-fill must leave every sentinel intact,
    and the gradients must equal those of the carved-from-one-allocation call."""
    from si_mamba_amd import _lib
    lib = _lib.load()
    b, d, L, N = 2, 24, 40, 16
    inp = scan_inputs(b, d, L, N, seed=31)
    t = {k: v.to(device) for k, v in inp.items()}
    out = torch.empty_like(t["u"])
    st = _lib.stream_ptr(device)
    rc = lib.simamba_selective_scan_fwd(_lib.ptr(t["u"]), _lib.ptr(t["delta"]), _lib.ptr(t["A"]), _lib.ptr(t["B"]),
                                        _lib.ptr(t["C"]), _lib.ptr(t["D"]), _lib.ptr(t["z"]),
                                        _lib.ptr(t["delta_bias"]), _lib.ptr(out), None, None, b, d, L, N, 0, 1,
                                        0, 0, 0, 0, 0, 0, st)
    assert rc == 0

    def bwd(acc):
        du, dd, dz = (torch.empty_like(t["u"]) for _ in range(3))
        rc = lib.simamba_selective_scan_bwd(
            _lib.ptr(t["u"]), _lib.ptr(t["delta"]), _lib.ptr(t["A"]), _lib.ptr(t["B"]), _lib.ptr(t["C"]),
            _lib.ptr(t["D"]), _lib.ptr(t["z"]), _lib.ptr(t["delta_bias"]), _lib.ptr(t["dout"]), None,
            _lib.ptr(du), _lib.ptr(dd), *[_lib.ptr(a) for a in acc[:4]], _lib.ptr(dz), _lib.ptr(acc[4]),
            b, d, L, N, 0, 1, 0, 0, 0, 0, 0, 0, st)
        assert rc == 0
        return du, dd, dz

    # (a) one slab, accumulators separated by 64-float sentinels, filled with garbage first
    sizes = [d * N, b * N * L, b * N * L, d, d]
    slab = torch.full((sum(sizes) + 64 * 6,), 7.0, device=device)
    acc, sent, o = [], [], 0
    for n in sizes:
        sent.append(slab[o:o + 64]); o += 64
        acc.append(slab[o:o + n]); o += n
    sent.append(slab[o:o + 64])
    g_a = bwd(acc)
    torch.cuda.synchronize()
    for s in sent:
        assert (s == 7.0).all()
    # (b) separately allocated small tensors with foreign live blocks in between
    acc_b, keep = [], []
    for n in sizes:
        acc_b.append(torch.full((n,), 3.0, device=device))
        keep.append(torch.full((32,), 5.0, device=device))
    g_b = bwd(acc_b)
    # (c) the product's carve
    acc_c = _lib.scan_bwd_accumulators(b, d, L, N, True, True, device)
    for a in acc_c:
        a.fill_(9.0)
    g_c = bwd([a.view(-1) for a in acc_c])
    torch.cuda.synchronize()
    for k in keep:
        assert (k == 5.0).all()
    for x, y, z in zip(acc, acc_b, acc_c):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(x, z.view(-1), rtol=1e-5, atol=1e-5)
    for x, y in zip(g_a, g_b):
        assert torch.equal(x, y)
