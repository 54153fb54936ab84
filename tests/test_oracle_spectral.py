"""CPU: spectral-ordering restatement -- the lower-triangle quirk, golden pins, assembly identities."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import spectral_ref as sr
from oracle.gen_golden import SPECTRAL_COMBOS, surface_centers, unit_ball_centers


def test_eigh_reads_lower_triangle_only():
    """SURVEY headline 4: eigh(I - D^-1 A) decomposes tril mirrored, not the RW Laplacian."""
    c = unit_ball_centers(2, 32, 3)
    adj = sr.create_graph_from_feature_space(c, 8, 10.0, True, False, True)
    L = sr.rw_laplacian(adj)
    assert not torch.allclose(L, L.transpose(1, 2))
    e1 = torch.linalg.eigvalsh(L)
    e2 = torch.linalg.eigvalsh(sr.eigh_lower(L))
    torch.testing.assert_close(e1, e2, rtol=0, atol=0)
    true_rw = torch.linalg.eigvals(L).real.sort(dim=1)[0]
    assert (true_rw - e1).abs().max() > 1e-4      # genuinely a different spectrum


def test_graph_properties():
    c = unit_ball_centers(3, 40, 4)
    adj = sr.create_graph_from_feature_space(c, 6, 10.0, False, False, True)
    assert torch.all(adj.sum(-1) == 6) and torch.all(torch.diagonal(adj, dim1=1, dim2=2) == 0)
    adj_s = sr.create_graph_from_feature_space(c, 6, 10.0, True, True, False)
    torch.testing.assert_close(adj_s, adj_s.transpose(1, 2))
    assert torch.all(torch.diagonal(adj_s, dim1=1, dim2=2) == 1.0)


@pytest.mark.parametrize("name,B,G,seed", [("spectral_g64", 4, 64, 0), ("spectral_g128", 2, 128, 1),
                                           ("spectral_g128_surface", 4, 128, 7)])
def test_spectral_golden_regression(name, B, G, seed):
    g = load_golden(name)
    # "_surface": FPS centres of clouds on thin closed surfaces (close eigenvalue pairs), the others Gaussian blobs
    centers = surface_centers(B, G, seed) if name.endswith("_surface") else unit_ball_centers(B, G, seed)
    np.testing.assert_array_equal(centers.numpy(), g["centers"])
    for cb in SPECTRAL_COMBOS:
        t = cb["tag"]
        adj = sr.create_graph_from_feature_space(centers, cb["knn"], cb["alpha"], cb["symmetric"],
                                                 cb["self_loop"], cb["binary"])
        np.testing.assert_array_equal(adj.numpy(), g[f"{t}.adj"])
        vals, vecs, all_vals, all_vecs = sr.calc_top_k_eigenvalues_eigenvectors(adj, 4, True)
        np.testing.assert_allclose(vals.numpy(), g[f"{t}.vals"], atol=1e-6)
        # the near-trivial eigenpair stays in the selection (lambda_0 ~ 0, either sign because of the quirk)
        assert vals[:, 0].abs().max() < 0.2
        res = torch.einsum("bij,bjk->bik", sr.eigh_lower(sr.rw_laplacian(adj)), vecs) - vecs * vals[:, None, :]
        assert res.abs().max() < 2e-5


def test_sast_assembly_equals_index_map():
    g = torch.Generator().manual_seed(0)
    tokens, pos = torch.randn(2, 16, 8, generator=g), torch.randn(2, 16, 8, generator=g)
    vecs = torch.randn(2, 16, 4, generator=g)
    x, p = sr.sast_assemble(tokens, pos, vecs, reverse=True)
    idx = sr.sast_index_map(sr.spectral_orders(vecs), reverse=True)
    assert x.shape == (2, 128, 8)
    torch.testing.assert_close(x, torch.gather(tokens, 1, idx[..., None].expand(-1, -1, 8)))
    torch.testing.assert_close(p, torch.gather(pos, 1, idx[..., None].expand(-1, -1, 8)))


def test_multilevel_travers_codes():
    v = torch.tensor([[[0.1, -1.0], [0.5, 1.0], [-0.3, 0.2], [0.9, -0.4]]])
    codes = sr.multilevel_travers(v, 2)
    assert codes.tolist() == [[0, 3, 1, 2]]


def test_hlt_assembly_overlapping_writes_as_index_map():
    """The reference's HLT block assembly (:1075-1112) overwrites blocks; its net effect is the index map the
    product uses (si_mamba_amd.spectral.hlt_index_map is pure index arithmetic, callable without a GPU)."""
    from si_mamba_amd.spectral import hlt_index_map
    g = torch.Generator().manual_seed(0)
    B, G, k = 2, 64, 3
    tokens, pos, center = torch.randn(B, G, 5, generator=g), torch.randn(B, G, 5, generator=g), torch.randn(B, G, 3, generator=g)
    vecs = torch.randn(B, G, k, generator=g)
    out_t, out_p, out_c, order = sr.hlt_order_and_assemble(tokens, pos, center, vecs, k, torch.rand(B, G, generator=g))
    slot = hlt_index_map(G, k)
    st = torch.gather(tokens, 1, order.unsqueeze(-1).expand(-1, -1, 5))
    want = torch.where((slot >= 0)[None, :, None], st[:, slot.clamp_min(0)], torch.zeros(()))
    assert torch.equal(out_t, want)
    nd, ng = G // 2 ** k, 2 ** k
    assert (slot[(nd + 2) * ng:] == -1).all() and (slot[:ng] == torch.arange(ng)).all()
