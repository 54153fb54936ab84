"""CPU: the oracle against fixtures produced by the REFERENCE'S OWN CODE (tests/golden/ref_*.npz).

oracle/pin_from_reference.py reads /root/reference/models/point_mamba.py and models/block.py at run time, compiles the
hot-path function bodies unmodified (device='cuda' mapped to the CPU) and stores their outputs; these tests hold
oracle/spectral_ref.py and the Block / MixerModel restatement to them.  Index work is compared bit for bit; eigenpairs
come out of the same stock torch.linalg.eigh call on the same matrix, so they are compared at 1e-6 (and are in fact
identical on the LAPACK build they were generated with).  The scan / conv arithmetic inside the mixer lives in the
absent mamba-ssm / causal-conv1d wheels: that half stays "parity unpinned" (oracle/__init__.py).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import pin_from_reference as pin
from oracle import scan_ref
from oracle import spectral_ref as sr
from oracle.gen_golden import SPECTRAL_COMBOS

NAMES = ["spectral_g64", "spectral_g128", "spectral_g128_surface"]


@pytest.fixture
def lapack_as_recorded():
    """eigh's last bits depend on the LAPACK thread count: run with the count the fixtures were generated with."""
    def setup(g):
        torch.set_num_threads(int(g["lapack_threads"]))
        return str(g["torch_version"]) == torch.__version__ and torch.get_num_threads() == int(g["lapack_threads"])
    before = torch.get_num_threads()
    yield setup
    torch.set_num_threads(before)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_graph_eigen_orders_equal_reference_outputs(name, lapack_as_recorded):
    g = load_golden("ref_" + name)
    same_lapack = lapack_as_recorded(g)
    centers = torch.from_numpy(g["centers"])
    exact_vecs = True
    for cb in SPECTRAL_COMBOS:
        t = cb["tag"]
        adj = sr.create_graph_from_feature_space(centers, cb["knn"], cb["alpha"], cb["symmetric"], cb["self_loop"],
                                                 cb["binary"])
        np.testing.assert_array_equal(adj.numpy(), g[f"{t}.adj"])                      # bit for bit
        vals, vecs, all_vals, _ = sr.calc_top_k_eigenvalues_eigenvectors(adj, 4, True)
        np.testing.assert_allclose(vals.numpy(), g[f"{t}.vals"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(all_vals.numpy(), g[f"{t}.all_vals"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(vecs.numpy(), g[f"{t}.vecs"], rtol=0, atol=1e-4)    # same sign: same LAPACK call
        exact_vecs &= np.array_equal(vecs.numpy(), g[f"{t}.vecs"]) and np.array_equal(vals.numpy(), g[f"{t}.vals"])
        # orders from the REFERENCE's eigenvectors: index work, exact
        rvecs = torch.from_numpy(g[f"{t}.vecs"])
        order = sr.spectral_orders(rvecs)
        np.testing.assert_array_equal(order.numpy(), g[f"{t}.order"])
        np.testing.assert_array_equal(sr.sast_index_map(order, reverse=True).numpy(), g[f"{t}.sast_index"])
        # and the token assembly itself (models/point_mamba.py:889-898, :982-989) on index-coded tokens
        B, G = centers.shape[:2]
        tok = torch.arange(G, dtype=torch.float32)[None, :, None].expand(B, G, 8).contiguous()
        x, p = sr.sast_assemble(tok, tok + 1000.0, rvecs, reverse=True)
        np.testing.assert_array_equal(x[:, :, 0].long().numpy(), g[f"{t}.sast_index"])
        np.testing.assert_array_equal(p[:, :, 0].long().numpy(), g[f"{t}.sast_index"] + 1000)
    # the same torch build with the same LAPACK thread count returns the reference's eigenpairs bit for bit
    if same_lapack:
        assert exact_vecs, "eigenpairs differ from the reference's on the LAPACK configuration that produced them"
    adj = torch.from_numpy(g["hardest.adj"])
    v, e, _, _ = sr.calc_top_k_eigenvalues_eigenvectors_symmetric(adj, 4, True)
    np.testing.assert_allclose(v.numpy(), g["hardest.sym.vals"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(e.numpy(), g["hardest.sym.vecs"], rtol=0, atol=1e-4)
    v, e, _, _ = sr.calc_top_k_eigenvalues_eigenvectors(adj, 4, False)
    np.testing.assert_allclose(v.numpy(), g["hardest.largest.vals"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(e.numpy(), g["hardest.largest.vecs"], rtol=0, atol=1e-4)
    if same_lapack:
        np.testing.assert_array_equal(e.numpy(), g["hardest.largest.vecs"])
    np.testing.assert_array_equal(sr.create_graph_from_centers(centers, 10, 0.0, True, True, False).numpy(),
                                  g["sigma_mean.adj"])
    np.testing.assert_array_equal(sr.create_graph_from_centers(centers, 10, 10.0, True, True, False).numpy(),
                                  g["centers_graph.adj"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_hlt_equals_reference_outputs(name, lapack_as_recorded):
    """HLT ordering + overlapping block assembly (models/point_mamba.py:1059-1112) with the reference's own torch.rand
    tie-break redrawn from the recorded seed."""
    g = load_golden("ref_" + name)
    lapack_as_recorded(g)
    centers = torch.from_numpy(g["centers"])
    vecs = torch.from_numpy(g["hlt.vecs"])
    B, G = centers.shape[:2]
    np.testing.assert_array_equal(sr.multilevel_travers(vecs, 3).numpy(), g["hlt.codes"])
    torch.manual_seed(int(g["hlt.rand_seed"]))
    rand = torch.rand(B, G)
    tok = (1.0 + torch.arange(G, dtype=torch.float32))[None, :, None].expand(B, G, 4).contiguous()
    out_t, out_p, out_c, order = sr.hlt_order_and_assemble(tok, tok + 1000.0, centers, vecs, 3, rand)
    np.testing.assert_array_equal(order.numpy(), g["hlt.order"])
    np.testing.assert_array_equal(out_t[:, :, 0].numpy(), g["hlt.tokens_index"])
    np.testing.assert_array_equal(out_p[:, :, 0].numpy(), g["hlt.pos_index"])
    np.testing.assert_array_equal(out_c.numpy(), g["hlt.center"])
    # the product's slot map is the same index arithmetic (callable without a GPU)
    from si_mamba_amd.spectral import hlt_index_map
    slot = hlt_index_map(G, 3)
    want = torch.from_numpy(g["hlt.tokens_index"])
    st = torch.gather(tok[:, :, 0], 1, order)
    got = torch.where(slot[None] >= 0, st[:, slot.clamp_min(0)], torch.zeros(()))
    np.testing.assert_array_equal(got.numpy(), want.numpy())


def _stack_from_fixture(g):
    """The oracle's restatement of the block stack (Add -> LayerNorm -> mixer, final norm) on the fixture's weights."""
    d, n_layer, B, L = (int(v) for v in g["dims"])
    sd = {k[len("param."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")}
    mixers = []
    for i in range(n_layer):
        m = scan_ref.MambaRef(d, layer_idx=i)
        m.load_state_dict({k[len(f"layers.{i}.mixer."):]: v for k, v in sd.items() if k.startswith(f"layers.{i}.mixer.")})
        mixers.append(m)
    ln = torch.nn.functional.layer_norm

    def norm(prefix, t):
        return ln(t, (d,), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)

    def run(x, pos):
        h, res = x + pos, None
        for i, m in enumerate(mixers):
            res = h if res is None else h + res
            h = m(norm(f"layers.{i}.norm", res))
        return norm("norm_f", h + res)
    return run, mixers, sd, (d, n_layer, B, L)


def test_oracle_block_stack_equals_reference_block_and_mixermodel():
    g = load_golden("ref_stack")
    run, mixers, sd, (d, n_layer, B, L) = _stack_from_fixture(g)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    pos = torch.from_numpy(g["pos"]).requires_grad_(True)
    out = run(x, pos)
    out.backward(torch.from_numpy(g["dout"]))
    # same torch ops in the same order as models/block.py:56-72 and models/point_mamba.py:247-258: exact
    np.testing.assert_array_equal(out.detach().numpy(), g["out"])
    np.testing.assert_array_equal(x.grad.numpy(), g["grad_x"])
    np.testing.assert_array_equal(pos.grad.numpy(), g["grad_pos"])
    for i, m in enumerate(mixers):
        for k, p in m.named_parameters():
            np.testing.assert_allclose(p.grad.numpy(), g[f"grad.layers.{i}.mixer.{k}"], rtol=1e-6, atol=1e-7)
    # one Block, both call forms: (mixer(LN(h)), h) and (mixer(LN(h + r)), h + r)
    h, r = torch.from_numpy(g["block.h"]), torch.from_numpy(g["block.r"])
    ln = torch.nn.functional.layer_norm
    w, b = sd["layers.0.norm.weight"], sd["layers.0.norm.bias"]
    np.testing.assert_array_equal(mixers[0](ln(h, (d,), w, b, 1e-5)).detach().numpy(), g["block.first.h"])
    np.testing.assert_array_equal(h.numpy(), g["block.first.r"])
    np.testing.assert_array_equal(mixers[0](ln(h + r, (d,), w, b, 1e-5)).detach().numpy(), g["block.next.h"])
    np.testing.assert_array_equal((h + r).numpy(), g["block.next.r"])


def test_reference_init_contract_and_state_dict_names():
    """What the reference's _init_weights / create_block leave behind (models/point_mamba.py:115-175), from its own
    run: parameter names as the state-dict contract has them, zero Linear biases except dt_proj.bias (_no_reinit),
    out_proj.weight within kaiming_uniform(a = sqrt 5) / sqrt(n_layer); the product's MixerModel builds the same."""
    g = load_golden("ref_stack")
    d, n_layer, _, _ = (int(v) for v in g["dims"])
    names = [str(n) for n in g["param_names"]]
    per_layer = ["mixer.A_log", "mixer.D", "mixer.in_proj.weight", "mixer.conv1d.weight", "mixer.conv1d.bias",
                 "mixer.x_proj.weight", "mixer.dt_proj.weight", "mixer.dt_proj.bias", "mixer.out_proj.weight",
                 "norm.weight", "norm.bias"]
    want = [f"layers.{i}.{k}" for i in range(n_layer) for k in per_layer] + ["norm_f.weight", "norm_f.bias"]
    assert names == want
    bound = float(g["init.out_proj_bound"])
    assert (g["init.out_proj_absmax"] <= bound * (1 + 1e-6)).all() and (g["init.out_proj_absmax"] > 0.9 * bound).all()
    for i in range(n_layer):
        assert np.abs(g[f"param.layers.{i}.mixer.dt_proj.bias"]).min() > 0          # survived the bias reset
    from si_mamba_amd.block import MixerModel
    torch.manual_seed(7)
    mine = MixerModel(d_model=d, n_layer=n_layer, rms_norm=False, drop_path=0.0)
    assert [k for k, _ in mine.state_dict().items()] == names
    for k, v in mine.state_dict().items():
        assert tuple(v.shape) == g["param." + k].shape, k
    for i in range(n_layer):
        w = mine.layers[i].mixer.out_proj.weight.detach().abs().max().item()
        assert 0.9 * bound < w <= bound * (1 + 1e-6)


@pytest.mark.skipif(not pin.reference_present(), reason="/root/reference is not on this machine")
def test_committed_fixtures_are_what_the_reference_returns_today():
    """Regenerates one fixture from the reference's files and compares it with the committed one."""
    methods, sast, hlt_fn, scope = pin.load_reference()
    import os
    import tempfile
    torch.set_num_threads(int(load_golden("ref_spectral_g64")["lapack_threads"]))
    old = pin.OUT
    with tempfile.TemporaryDirectory() as tmp:
        # the generator reads the centres from the committed spectral_g64.npz and writes ref_spectral_g64.npz
        os.symlink(os.path.join(old, "spectral_g64.npz"), os.path.join(tmp, "spectral_g64.npz"))
        pin.OUT = tmp
        try:
            rec = pin.spectral_fixture("spectral_g64", methods, sast, hlt_fn)
        finally:
            pin.OUT = old
    g = load_golden("ref_spectral_g64")
    assert set(rec) == set(g)
    for k in g:
        np.testing.assert_array_equal(np.asarray(rec[k]), g[k], err_msg=k)


@pytest.mark.skipif(not pin.reference_present(), reason="/root/reference is not on this machine")
def test_reference_create_block_over_the_shim_has_the_contract_names():
    """The reference's own create_block (models/point_mamba.py:147-175) and Block (models/block.py:17-45), compiled
    from its files, building THIS package's Mamba through the import shim: constructs without a GPU, and its
    state_dict() carries the names and shapes of the reference's training log (tests/golden/param_table_*.json)."""
    import ast
    import json
    import math
    import os
    from functools import partial
    from typing import Optional
    import si_mamba_amd
    si_mamba_amd.install_shim()
    from mamba_ssm.modules.mamba_simple import Mamba          # the reference's import line, models/point_mamba.py:25
    scope = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "math": math, "partial": partial,
             "Tensor": torch.Tensor, "Optional": Optional, "Mamba": Mamba, "DropPath": pin._DropPath, "RMSNorm": None,
             "layer_norm_fn": None, "rms_norm_fn": None}
    pin._exec_defs([pin._find(pin._parse(pin.REF_BLOCK).body, ast.ClassDef, "Block")], scope, pin.REF_BLOCK)
    tree = pin._parse(pin.REF_MODEL)
    pin._exec_defs([pin._find(tree.body, ast.FunctionDef, "_init_weights"),
                    pin._find(tree.body, ast.FunctionDef, "create_block")], scope, pin.REF_MODEL)
    blk = scope["create_block"](384, layer_idx=0, drop_path=0.0)
    blk.apply(partial(scope["_init_weights"], n_layer=12))
    assert type(blk.mixer) is Mamba and blk.layer_idx == 0 and blk.mixer.layer_idx == 0
    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "param_table_finetune_hardest.json")))
    want = {r["name"][len("blocks.layers.0."):]: tuple(r["shape"]) for r in table["params"]
            if r["name"].startswith("blocks.layers.0.")}
    got = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    assert got == want
    assert getattr(blk.mixer.dt_proj.bias, "_no_reinit", False)
    assert blk.mixer.dt_proj.bias.detach().abs().min() > 0
