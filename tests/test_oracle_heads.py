"""CPU: the oracle restatements of the segmentation interpolation and the MAE Chamfer loss reproduce their committed
fixtures (regression pins made by oracle/gen_golden.py; parity unpinned against the reference, which holds none),
and the index arithmetic of si_mamba_amd.mae equals the reference-style boolean-mask enumeration."""
import os

import numpy as np
import torch

from oracle import mae_ref, seg_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_interp_fixture():
    z = np.load(os.path.join(GOLD, "interp_seg.npz"))
    out, idx, w = seg_ref.three_nn_interpolate(torch.from_numpy(z["xyz1"]), torch.from_numpy(z["xyz2"]),
                                               torch.from_numpy(z["feats"]))
    assert np.array_equal(idx.numpy(), z["idx"])
    np.testing.assert_allclose(w.numpy(), z["weight"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(w.sum(-1).numpy(), 1.0, rtol=1e-5)


def test_chamfer_fixture_and_definition():
    z = np.load(os.path.join(GOLD, "chamfer_mae.npz"))
    x = torch.from_numpy(z["pred"]).requires_grad_(True)
    y = torch.from_numpy(z["gt"])
    d = mae_ref.chamfer_distance(x, y)
    (d * torch.from_numpy(z["wsum"])).sum().backward()
    np.testing.assert_allclose(d.detach().numpy(), z["dist"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(x.grad.numpy(), z["grad_pred"], rtol=1e-5, atol=1e-7)
    # definition check on one pair, by loops
    p, g = z["pred"][3], z["gt"][3]
    a = np.mean([min(((pi - gj) ** 2).sum() for gj in g) for pi in p])
    b = np.mean([min(((pi - gj) ** 2).sum() for pi in p) for gj in g])
    assert abs((a + b) - z["dist"][3]) < 1e-5


def test_mae_index_arithmetic_matches_boolean_mask_enumeration():
    from si_mamba_amd.mae import masked_positions, sequence_positions
    g = torch.Generator().manual_seed(0)
    B, k, G, nm = 3, 4, 16, 9
    orders = torch.stack([torch.stack([torch.randperm(G, generator=g) for _ in range(k)]) for _ in range(B)])
    mask = torch.zeros(B, G, dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(G, generator=g)[:nm]] = True
    vis_pos, msk_pos, smask = masked_positions(orders, mask)
    # reference form: P @ mask, lists per ordering, reversed copy (models/point_mamba.py:2464, :2527-2528, :3192-3193)
    P = mae_ref.permutation_matrices(orders, G)
    sm = torch.matmul(P, mask.unsqueeze(1).unsqueeze(-1).float()).squeeze(-1).bool()
    assert torch.equal(sm, smask)
    final = torch.cat((torch.cat(list(torch.unbind(sm, 1)), -1), torch.cat(list(torch.unbind(sm, 1)), -1).flip(-1)), 1)
    want_m = torch.stack([torch.nonzero(final[b]).squeeze(-1) for b in range(B)])
    want_v = torch.stack([torch.nonzero(~final[b]).squeeze(-1) for b in range(B)])
    assert torch.equal(sequence_positions(msk_pos, G, True), want_m)
    assert torch.equal(sequence_positions(vis_pos, G, True), want_v)
