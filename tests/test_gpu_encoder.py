"""GPU parity of the patch-encoder streaming kernels (csrc/bn_relu.hip) against the stock torch ops the
reference's Encoder is made of (models/point_mamba.py:42-73): nn.BatchNorm1d + ReLU, torch.max over the
points of a patch, and the concat of the global feature.  Tolerances: 1e-4 fp32 / 2e-2 bf16."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def nerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("rows,C,group,dtype", [(4096, 128, 0, torch.float32), (2048, 512, 32, torch.float32),
                                               (1000, 96, 0, torch.float32), (768, 384, 16, torch.float32),
                                               (1024, 1536, 0, torch.float32), (2048, 1280, 512, torch.float32),
                                               (4096, 256, 32, torch.bfloat16)])
def test_bn_relu_matches_torch(rows, C, group, dtype, device):
    from si_mamba_amd.encoder_ops import bn_relu_fn
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.7).to(device).to(dtype)
    gt = (torch.randn(rows // group, C, generator=g).to(device) if group else None)
    dy = torch.randn(rows, C, generator=g).to(device).to(dtype)
    bn_a, bn_b = nn.BatchNorm1d(C).to(device), nn.BatchNorm1d(C).to(device)
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(C, generator=g) + 0.5); bn_a.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn_b.load_state_dict(bn_a.state_dict())
    xa = x.clone().requires_grad_(True)
    ga = None if gt is None else gt.clone().requires_grad_(True)
    ya = bn_relu_fn(xa, bn_a, gterm=ga, group=group)
    ya.backward(dy)
    xb = x.float().clone().requires_grad_(True)
    gb = None if gt is None else gt.clone().requires_grad_(True)
    xe = xb if gb is None else xb + gb.repeat_interleave(group, dim=0)
    yb = torch.relu(bn_b(xe))
    yb.backward(dy.float())
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert nerr(ya, yb) < tol
    assert nerr(xa.grad, xb.grad) < tol * 5
    assert nerr(bn_a.weight.grad, bn_b.weight.grad) < tol * 5
    assert nerr(bn_a.bias.grad, bn_b.bias.grad) < tol * 5
    if gt is not None:
        assert nerr(ga.grad, gb.grad) < tol * 5
    # buffers follow nn.BatchNorm1d
    assert nerr(bn_a.running_mean, bn_b.running_mean) < tol
    assert nerr(bn_a.running_var, bn_b.running_var) < tol
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1
    # eval mode uses them
    bn_a.eval(); bn_b.eval()
    with torch.no_grad():
        ye = bn_relu_fn(x, bn_a, gterm=gt, group=group)
        yr = torch.relu(bn_b(x.float() if gt is None else x.float() + gt.repeat_interleave(group, dim=0)))
    assert nerr(ye, yr) < tol


def test_bn_relu_large_mean_is_stable(device):
    """Shifted sums: a channel mean 1000x its spread must not cancel."""
    from si_mamba_amd.encoder_ops import bn_relu_fn
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(8192, 64, generator=g) * 0.01 + 10.0).to(device)
    bn = nn.BatchNorm1d(64).to(device)
    y = bn_relu_fn(x, bn)
    ref = torch.relu(torch.nn.functional.batch_norm(x.double(), None, None, bn.weight.double(), bn.bias.double(), True))
    assert nerr(y, ref) < 2e-3


@pytest.mark.parametrize("groups,n,C,dtype", [(512, 32, 256, torch.float32), (100, 17, 384, torch.float32),
                                              (256, 32, 384, torch.bfloat16)])
def test_group_max_matches_torch(groups, n, C, dtype, device):
    from si_mamba_amd.encoder_ops import group_max_fn
    g = torch.Generator().manual_seed(groups)
    x = torch.randn(groups, n, C, generator=g).to(device).to(dtype)
    x[0, 3, :8] = x[0, 5, :8] = 9.0                                  # a tie: the first maximum takes the gradient
    dy = torch.randn(groups, C, generator=g).to(device).to(dtype)
    xa = x.clone().requires_grad_(True)
    ya = group_max_fn(xa)
    ya.backward(dy)
    assert torch.equal(ya, x.max(dim=1)[0])
    want = torch.zeros_like(x)
    idx = x.float().cpu().numpy().argmax(axis=1)                       # numpy argmax: first maximum
    want.scatter_(1, torch.from_numpy(idx).to(device).unsqueeze(1), dy.unsqueeze(1))
    assert torch.equal(xa.grad, want)


@pytest.mark.parametrize("train", [True, False])
def test_encoder_fused_matches_composed(train, device):
    """Same module, fused HIP path vs the composed torch path (the reference's op sequence)."""
    from si_mamba_amd.point_mamba import Encoder
    torch.manual_seed(0)
    enc_a, enc_b = Encoder(384).to(device), Encoder(384).to(device)
    enc_b.load_state_dict(enc_a.state_dict())
    enc_b.fused = False
    enc_a.train(train); enc_b.train(train)
    pts = torch.randn(4, 16, 32, 3, device=device)
    dy = torch.randn(4, 16, 384, device=device)
    pa, pb = pts.clone().requires_grad_(True), pts.clone().requires_grad_(True)
    ya, yb = enc_a(pa), enc_b(pb)
    assert nerr(ya, yb) < 1e-4
    if train:
        ya.backward(dy); yb.backward(dy)
        assert nerr(pa.grad, pb.grad) < 2e-3
        for (k, a), (_, b) in zip(enc_a.named_parameters(), enc_b.named_parameters()):
            if k in ("first_conv.0.bias", "first_conv.3.bias", "second_conv.0.bias"):
                # biases in front of a BatchNorm (first_conv.3.bias reaches BN2 through both concat halves): the
                # gradient is exactly 0, both sides hold rounding noise
                assert float(a.grad.abs().max()) < 1e-4 and float(b.grad.abs().max()) < 1e-4, k
            else:
                assert nerr(a.grad, b.grad) < 2e-3, k
        for (k, a), (_, b) in zip(enc_a.named_buffers(), enc_b.named_buffers()):
            assert nerr(a, b) < 1e-4, k


def test_bn_relu_argument_errors(device):
    from si_mamba_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 8, device=device)
    m = torch.zeros(8, device=device)
    part = torch.zeros(1, 2, 8, device=device)
    # group must divide 256 and rows
    # group must divide rows
    rc = lib.simamba_bn_relu_fwd(x.data_ptr(), m.data_ptr(), 7, None, None, None, None, 0.1, 1e-5, 1, x.data_ptr(),
                                 m.data_ptr(), m.data_ptr(), part.data_ptr(), 64, 8, 0, 0, None)
    assert rc < 0
    # C % 4, and row stride >= C
    rc = lib.simamba_bn_relu_fwd(x.data_ptr(), None, 0, None, None, None, None, 0.1, 1e-5, 1, x.data_ptr(),
                                 m.data_ptr(), m.data_ptr(), part.data_ptr(), 64, 6, 0, 0, None)
    assert rc < 0
    rc = lib.simamba_bn_relu_fwd(x.data_ptr(), None, 0, None, None, None, None, 0.1, 1e-5, 1, x.data_ptr(),
                                 m.data_ptr(), m.data_ptr(), part.data_ptr(), 64, 8, 4, 0, None)
    assert rc < 0


@pytest.mark.parametrize("rows,cin,cout,bias", [(262144, 256, 512, False), (65536, 128, 256, True), (32768, 3, 128, True),
                                                (1000, 64, 32, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_token_linear_matches_linear(rows, cin, cout, bias, dtype, device):
    """token_linear: F.linear's forward, and a weight gradient formed as a split-K batched product over 64 row slabs
    (fp32 partials) instead of one K = rows GEMM -- same gradients as autograd's F.linear (to the rounding of a
    different summation order; under autocast the reference path rounds dW to bf16, this one does not).
    (1000 rows: below the slab size, the plain product.)"""
    from si_mamba_amd.encoder_ops import token_linear
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g).to(device)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(device)
    b = torch.randn(cout, generator=g).to(device) if bias else None
    dy = torch.randn(rows, cout, generator=g).to(device)
    grads = {}
    for name, fn in (("ours", token_linear), ("torch", torch.nn.functional.linear)):
        xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ba = None if b is None else b.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == torch.bfloat16)):
            y = fn(xa, wa, ba)
        assert y.dtype == dtype
        y.backward(dy.to(dtype))
        grads[name] = (y.detach(), xa.grad, wa.grad, None if ba is None else ba.grad)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert torch.equal(grads["ours"][0], grads["torch"][0])                    # the forward IS F.linear
    for got, want in zip(grads["ours"][1:], grads["torch"][1:]):
        if want is not None:
            assert got.dtype == want.dtype and nerr(got, want) < tol
    if dtype == torch.float32:                                                 # and against float64 for the weight gradient
        want = dy.double().t() @ x.double()
        assert nerr(grads["ours"][2], want) < 1e-5
