"""CPU: the scan / conv / mixer restatement against an independent closed form and the golden pins."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import scan_ref
from oracle.gen_golden import scan_inputs


@pytest.mark.parametrize("with_z,with_D,with_bias", [(True, True, True), (False, False, False)])
def test_scan_ref_matches_closed_form(with_z, with_D, with_bias):
    inp = scan_inputs(2, 6, 24, 4, seed=11, with_z=with_z, with_D=with_D, with_bias=with_bias)
    got = scan_ref.selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], inp["z"],
                                      inp["delta_bias"], delta_softplus=True, acc_dtype=torch.float64)
    want = scan_ref.selective_scan_closed_form(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"],
                                               inp["z"], inp["delta_bias"], delta_softplus=True)
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-5)


def test_scan_ref_grouped_bc_and_last_state():
    inp = scan_inputs(1, 4, 10, 3, seed=5)
    a = scan_ref.selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"])
    b, last = scan_ref.selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"][:, None], inp["C"][:, None],
                                          return_last_state=True)
    torch.testing.assert_close(a, b)
    assert last.shape == (1, 4, 3)


@pytest.mark.parametrize("name", ["scan_cfg1", "scan_l128", "scan_multichunk_ragged", "scan_odd"])
def test_scan_golden_regression(name):
    g = load_golden(name)
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    out, last = scan_ref.selective_scan_ref(t["u"], t["delta"], t["A"], t["B"], t["C"], t.get("D"), t.get("z"),
                                            t.get("delta_bias"), delta_softplus=True, return_last_state=True)
    torch.testing.assert_close(out, t["out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(last, t["last_state"], rtol=1e-5, atol=1e-6)


def test_conv_ref_matches_loop():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 9, generator=g)
    w = torch.randn(3, 4, generator=g)
    b = torch.randn(3, generator=g)
    got = scan_ref.causal_conv1d_ref(x, w, b, "silu")
    want = torch.zeros_like(x)
    for t in range(9):
        acc = b.clone()[None].repeat(2, 1)
        for k in range(4):
            s = t - 3 + k
            if s >= 0:
                acc = acc + w[:, k] * x[:, :, s]
        want[:, :, t] = acc * torch.sigmoid(acc)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["conv_cfg1", "conv_odd"])
def test_conv_golden_regression(name):
    g = load_golden(name)
    act = "silu" if int(g["silu"]) else None
    out = scan_ref.causal_conv1d_ref(torch.from_numpy(g["x"]), torch.from_numpy(g["w"]),
                                     torch.from_numpy(g["bias"]) if "bias" in g else None, act)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-6)


def test_mamba_ref_block_golden():
    g = load_golden("mamba_block_cfg1")
    m = scan_ref.MambaRef(128, layer_idx=0)
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")})
    out = m(torch.from_numpy(g["hidden"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    assert getattr(m.dt_proj.bias, "_no_reinit", False)
