"""GPU parity: causal depthwise conv1d fwd/bwd through the C ABI vs the CPU oracle (1e-3 fp32, 1e-2 bf16)."""
import pytest
import torch

from conftest import load_golden
from oracle import scan_ref

pytestmark = pytest.mark.gpu


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


@pytest.mark.parametrize("name", ["conv_cfg1", "conv_odd"])
def test_conv_golden(name, device):
    from si_mamba_amd import causal_conv1d_fn
    g = {k: torch.from_numpy(v) for k, v in load_golden(name).items()}
    act = "silu" if int(g["silu"]) else None
    x = g["x"].to(device).requires_grad_(True)
    w = g["w"].to(device).requires_grad_(True)
    b = g["bias"].to(device).requires_grad_(True) if "bias" in g else None
    out = causal_conv1d_fn(x, w, b, act)
    out.backward(g["dout"].to(device))
    assert nerr(out, g["out"]) < 1e-3
    assert nerr(x.grad, g["grad_x"]) < 1e-3
    assert nerr(w.grad, g["grad_w"]) < 1e-3
    if b is not None:
        assert nerr(b.grad, g["grad_bias"]) < 1e-3


@pytest.mark.parametrize("shape", [(2, 256, 64, 4), (3, 100, 128, 4), (2, 24, 1024, 4), (2, 33, 50, 3),
                                   (1, 7, 1, 2), (1, 16, 2100, 4), (64, 768, 128, 4),
                                   (1, 16, 4096, 4), (2, 8, 8, 4), (3, 40, 16, 3)])   # aligned rows: the pipelined backward
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [None, "silu"])
def test_conv_random(shape, dtype, act, device):
    from si_mamba_amd import causal_conv1d_fn
    b, d, L, W = shape
    g = torch.Generator().manual_seed(L + d)
    x = torch.randn(b, d, L, generator=g).to(dtype)
    w = torch.randn(d, W, generator=g) * 0.5
    bias = torch.randn(d, generator=g)
    dout = torch.randn(b, d, L, generator=g).to(dtype)
    xr, wr, br = (t.float().clone().requires_grad_(True) for t in (x, w, bias))
    want = scan_ref.causal_conv1d_ref(xr, wr, br, act)
    want.backward(dout.float())
    xd, wd, bd = (t.to(device).requires_grad_(True) for t in (x, w, bias))
    got = causal_conv1d_fn(xd, wd, bd, act)
    got.backward(dout.to(device))
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    assert got.dtype == dtype
    assert nerr(got, want) < tol
    assert nerr(xd.grad, xr.grad) < tol
    assert nerr(wd.grad, wr.grad) < tol * (4 if dtype == torch.bfloat16 else 1)
    assert nerr(bd.grad, br.grad) < tol * (4 if dtype == torch.bfloat16 else 1)


def test_conv_bad_activation_raises(device):
    from si_mamba_amd import causal_conv1d_fn
    with pytest.raises(NotImplementedError):
        causal_conv1d_fn(torch.zeros(1, 4, 8, device=device), torch.zeros(4, 4, device=device), None, "relu")
