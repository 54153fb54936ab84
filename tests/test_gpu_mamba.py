"""GPU parity: the drop-in Mamba mixer / Block / MixerModel vs the CPU restatement (BASELINE config 1)."""
import pytest
import torch

from conftest import load_golden
from oracle import scan_ref

pytestmark = pytest.mark.gpu


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


def test_mamba_block_cfg1_golden(device):
    """Single mixer, B=2, L=64, d_model=128, d_state=16: forward + every gradient vs the golden vector."""
    from si_mamba_amd import Mamba
    g = load_golden("mamba_block_cfg1")
    torch.backends.cuda.matmul.allow_tf32 = False
    m = Mamba(128, layer_idx=0).to(device)
    m.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("param.")})
    h = torch.from_numpy(g["hidden"]).to(device).requires_grad_(True)
    out = m(h)
    out.backward(torch.from_numpy(g["dout"]).to(device))
    assert nerr(out, torch.from_numpy(g["out"])) < 1e-3
    assert nerr(h.grad, torch.from_numpy(g["grad_hidden"])) < 1e-3
    for k, p in m.named_parameters():
        assert nerr(p.grad, torch.from_numpy(g["grad." + k])) < 1e-3, k


def test_block_and_stack_match_reference_semantics(device):
    """Block returns (mixer(LN(residual)), residual); MixerModel = x+pos, blocks, norm_f(h+residual)."""
    from si_mamba_amd.block import MixerModel
    torch.manual_seed(0)
    mm = MixerModel(d_model=64, n_layer=3, drop_path=0.).to(device).eval()
    ref = [scan_ref.MambaRef(64, layer_idx=i) for i in range(3)]
    for layer, r in zip(mm.layers, ref):
        r.load_state_dict({k: v.cpu() for k, v in layer.mixer.state_dict().items()})
    x, pos = torch.randn(2, 40, 64), torch.randn(2, 40, 64)
    got = mm(x.to(device), pos.to(device))
    h, res = x + pos, None
    for layer, r in zip(mm.layers, ref):
        res = h if res is None else h + res
        ln = torch.nn.functional.layer_norm(res, (64,), layer.norm.weight.cpu(), layer.norm.bias.cpu(), 1e-5)
        h = r(ln)
    want = torch.nn.functional.layer_norm(h + res, (64,), mm.norm_f.weight.cpu(), mm.norm_f.bias.cpu(), 1e-5)
    assert nerr(got, want) < 1e-3
    hs, r0 = mm.layers[0](x.to(device), None)
    assert torch.equal(r0.cpu(), x)


@pytest.mark.parametrize("shape", [(2, 50, 64), (2, 1024, 384), (4, 208, 384)])
def test_mamba_bf16_autocast(shape, device):
    """The mixer under bf16 autocast (tools/runner_pretrain.py:243) against the oracle restated with the autocast
    roundings (MambaRef.forward io_dtype: every GEMM, the conv and the scan read and write bf16, accumulate fp32;
    conv weights, A, D, dt bias fp32).  north_star tolerance for bf16: 1e-2 on the normalised error -- forward and
    the gradients of the input and of every parameter (the oracle's backward carries fp32 gradients between the ops
    where the device carries bf16 ones: that difference is inside the same budget)."""
    from si_mamba_amd import Mamba
    B, L, d = shape
    torch.manual_seed(1)
    m = Mamba(d).to(device)
    ref = scan_ref.MambaRef(d)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    h = torch.randn(B, L, d)
    dout = torch.randn(B, L, d)
    hd = h.to(device).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got = m(hd)
    assert got.dtype == torch.bfloat16
    got.backward(dout.to(device).to(got.dtype))
    hr = h.clone().requires_grad_(True)
    want = ref(hr, io_dtype=torch.bfloat16)
    want.backward(dout.to(torch.bfloat16).float())
    assert nerr(got.float(), want) < 1e-2
    assert nerr(hd.grad, hr.grad) < 1e-2
    pg = dict(m.named_parameters())
    for k, p in ref.named_parameters():
        assert nerr(pg[k].grad, p.grad) < 1e-2, k


def test_inference_params_refused(device):
    from si_mamba_amd import Mamba
    with pytest.raises(NotImplementedError):
        Mamba(32).to(device)(torch.zeros(1, 4, 32, device=device), inference_params=object())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 128), (3, 200, 96), (2, 1024, 64)])
def test_fused_inner_fn_equals_composed_ops(shape, dtype, device):
    """mamba_inner_fn (strided, copy-free) against the same mixer built from the separate ops, and both
    against the oracle in fp32: forward, input gradient and every parameter gradient."""
    from si_mamba_amd import Mamba
    B, L, d = shape
    torch.manual_seed(3)
    fast = Mamba(d, use_fast_path=True).to(device)
    slow = Mamba(d, use_fast_path=False).to(device)
    slow.load_state_dict(fast.state_dict())
    h = torch.randn(B, L, d)
    dout = torch.randn(B, L, d)
    outs = []
    for m in (fast, slow):
        hd = h.to(device).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == torch.bfloat16)):
            o = m(hd)
        o.backward(dout.to(device).to(o.dtype))
        outs.append((o.float(), hd.grad, {k: p.grad for k, p in m.named_parameters()}))
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    assert nerr(outs[0][0], outs[1][0]) < tol
    assert nerr(outs[0][1], outs[1][1]) < tol
    for k in outs[0][2]:
        assert nerr(outs[0][2][k], outs[1][2][k]) < tol, k
    if dtype == torch.float32:
        ref = scan_ref.MambaRef(d)
        ref.load_state_dict({k: v.cpu() for k, v in fast.state_dict().items()})
        hr = h.clone().requires_grad_(True)
        ro = ref(hr)
        ro.backward(dout)
        assert nerr(outs[0][0], ro) < 1e-3
        assert nerr(outs[0][1], hr.grad) < 1e-3
        for k, p in ref.named_parameters():
            assert nerr(outs[0][2][k], p.grad) < 1e-3, k
