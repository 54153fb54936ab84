"""GPU parity: fused DropPath-scaled add + LayerNorm vs the reference's composed torch form
(models/block.py:56-60), forward and every gradient; 1e-3 fp32 / 1e-2 bf16."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def nerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


def composed(hidden, residual, w, b, eps, rowscale):
    h = hidden.float()
    if residual is not None:
        if rowscale is not None:
            h = h * rowscale[:, None, None]
        res = h + residual
    else:
        res = h
    return torch.nn.functional.layer_norm(res, (res.shape[-1],), w, b, eps), res


@pytest.mark.parametrize("shape", [(64, 1024, 384), (3, 50, 128), (2, 7, 2048), (5, 1, 36), (1, 300, 512)])
@pytest.mark.parametrize("has_res,has_scale", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("hdtype", [torch.float32, torch.bfloat16])
def test_add_layer_norm_matches_composed(shape, has_res, has_scale, hdtype, device):
    from si_mamba_amd.add_norm import add_layer_norm_fn
    B, L, d = shape
    g = torch.Generator().manual_seed(B * 1000 + d)
    hidden = torch.randn(B, L, d, generator=g).to(hdtype)
    residual = torch.randn(B, L, d, generator=g) if has_res else None
    w = 1.0 + 0.1 * torch.randn(d, generator=g)
    b = 0.1 * torch.randn(d, generator=g)
    scale = (torch.rand(B, generator=g) > 0.3).float() / 0.7 if has_scale else None
    dn = torch.randn(B, L, d, generator=g)
    dr = torch.randn(B, L, d, generator=g)

    ref_in = [t.clone().float().requires_grad_(True) if t is not None else None for t in (hidden, residual, w, b)]
    wn, wr = composed(ref_in[0], ref_in[1], ref_in[2], ref_in[3], 1e-5, scale)
    (wn * dn).sum().add((wr * dr).sum()).backward()

    dev_in = [t.to(device).clone().requires_grad_(True) if t is not None else None for t in (hidden, residual, w, b)]
    out_dtype = torch.float32 if hdtype == torch.float32 else torch.bfloat16
    gn, gr = add_layer_norm_fn(dev_in[0], dev_in[1], dev_in[2], dev_in[3], 1e-5,
                               rowscale=None if scale is None else scale.to(device), out_dtype=out_dtype)
    assert gn.dtype == out_dtype and gr.dtype == torch.float32
    ((gn.float() * dn.to(device)).sum() + (gr * dr.to(device)).sum()).backward()
    tol = 1e-3 if hdtype == torch.float32 else 1e-2
    assert nerr(gn, wn) < tol and nerr(gr, wr) < tol
    assert nerr(dev_in[0].grad, ref_in[0].grad) < tol
    if has_res:
        assert nerr(dev_in[1].grad, ref_in[1].grad) < tol
    assert nerr(dev_in[2].grad, ref_in[2].grad) < tol * 4
    assert nerr(dev_in[3].grad, ref_in[3].grad) < tol * 4


def test_block_training_with_drop_path_matches_composed(device):
    """Block.forward in training mode (DropPath active): same random keep-mask, fused vs composed."""
    from si_mamba_amd.block import create_block
    torch.manual_seed(0)
    blk = create_block(64, layer_idx=0, drop_path=0.5).to(device).train()
    x, res = torch.randn(8, 40, 64, device=device), torch.randn(8, 40, 64, device=device)
    torch.manual_seed(123)
    h1, r1 = blk(x, res)
    torch.manual_seed(123)
    mask = blk.drop_path.rowscale(x)
    r2 = x * mask[:, None, None] + res
    h2 = blk.mixer(blk.norm(r2))
    assert nerr(r1, r2) < 1e-5 and nerr(h1, h2) < 1e-4
    assert (mask == 0).any() and (mask > 1).any()
