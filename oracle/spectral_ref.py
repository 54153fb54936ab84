"""CPU restatement of the reference's spectral token ordering.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, function by function,
reference models/point_mamba.py:
  create_graph_from_centers                                  :620-661
  create_graph_from_feature_space_gpu_weighted_adjacency     :664-715
  calc_top_k_eigenvalues_eigenvectors                        :717-761
  calc_top_k_eigenvalues_eigenvectors_symmetric              :764-814
  sort_points_by_fiedler                                     :817-826
  multilevel_travers                                         :829-841
  SAST token assembly                                        :889-898, :982-989
with the hard-coded ``.cuda()`` moves dropped.  The eigensolve is the same
stock ``torch.linalg.eigh`` call on the *unsymmetrised* ``I - D^-1 A``: LAPACK
reads the lower triangle only, so the matrix actually decomposed is
``tril(L) + tril(L,-1)^T`` (SURVEY.md headline 4).

The reference holds no fixtures for these functions: parity unpinned, but
every library call below is the one the reference itself makes.
"""
from __future__ import annotations

import torch


def _knn(points, k, self_loop):
    # :682-694 -- distances by explicit broadcast difference, top-(k+1) of -dist
    diff = points.unsqueeze(2) - points.unsqueeze(1)
    dist = torch.sqrt(torch.sum(diff ** 2, dim=-1))
    neg, idx = torch.topk(-dist, k=k + 1, largest=True, dim=-1)
    d = -neg
    if not self_loop:
        idx, d = idx[:, :, 1:], d[..., 1:]
    return dist, d, idx


def _scatter_adjacency(B, N, idx, w, symmetric, binary, dtype):
    adj = torch.zeros(B, N, N, dtype=dtype)
    b_idx = torch.arange(B)[:, None, None]
    n_idx = torch.arange(N)[:, None]
    val = 1.0 if binary else w
    adj[b_idx, n_idx, idx] = val
    if symmetric:
        adj[b_idx, idx, n_idx] = val
    return adj


def create_graph_from_feature_space(points, k=5, alpha=1, symmetric=False,
                                    self_loop=False, binary=False):
    """reference :664-715 (the graph SAST uses, :872)."""
    B, N, _ = points.shape
    _, d, idx = _knn(points, k, self_loop)
    w = torch.exp((-1) * alpha * d ** 2)
    return _scatter_adjacency(B, N, idx, w, symmetric, binary, points.dtype)


def create_graph_from_centers(points, k=5, alpha=1, symmetric=False, self_loop=False,
                              binary=False, self_alpha=None):
    """reference :620-661.  ``self_alpha`` is the module attribute the reference
    tests against 0 (:647); the *argument* alpha is what it uses otherwise."""
    B, N, _ = points.shape
    dist, d, idx = _knn(points, k, self_loop)
    sigma = torch.mean(dist)
    if (alpha if self_alpha is None else self_alpha) == 0:
        w = torch.exp(-d ** 2 / (2 * sigma ** 2))
    else:
        w = torch.exp((-1) * alpha * d ** 2)
    return _scatter_adjacency(B, N, idx, w, symmetric, binary, points.dtype)


def rw_laplacian(adj):
    """reference :727-740, batched: (A+A^T)/2, deg=rowsum, L = I - diag(1/(deg+1e-6)) A."""
    A = (adj + adj.transpose(1, 2)) / 2
    deg = A.sum(dim=2)
    dinv = 1.0 / (deg + 1e-6)
    eye = torch.eye(A.shape[1], dtype=A.dtype)
    return eye[None] - dinv[:, :, None] * A


def sym_laplacian(adj):
    """reference :776-790: L = I - D^-1/2 A D^-1/2 (no epsilon)."""
    A = (adj + adj.transpose(1, 2)) / 2
    deg = A.sum(dim=2)
    dis = torch.pow(deg, -0.5)
    eye = torch.eye(A.shape[1], dtype=A.dtype)
    return eye[None] - dis[:, :, None] * A * dis[:, None, :]


def eigh_lower(Lm):
    """What eigh(UPLO='L') decomposes: tril(L) mirrored."""
    lo = torch.tril(Lm)
    return lo + torch.tril(Lm, -1).transpose(-1, -2)


def calc_top_k_eigenvalues_eigenvectors(adj, k, smallest):
    """reference :717-761 -- per-sample loop kept on purpose (it is the reference's)."""
    B, N, _ = adj.shape
    vals = torch.zeros(B, k)
    vecs = torch.zeros(B, N, k)
    all_vals = torch.zeros(B, N)
    all_vecs = torch.zeros(B, N, N)
    for i in range(B):
        A = adj[i]
        A = (A + A.t()) / 2
        Dm = torch.diag(torch.sum(A, dim=1))
        D_inv = torch.diag(1.0 / (torch.diag(Dm) + 1e-6))
        Lrw = torch.eye(N) - torch.matmul(D_inv, A)
        ev, evec = torch.linalg.eigh(Lrw)
        tv, ti = torch.topk(ev, k, largest=not smallest, sorted=True)
        vals[i] = tv
        vecs[i] = evec[:, ti]
        all_vals[i] = ev
        all_vecs[i] = evec
    return vals, vecs, all_vals, all_vecs


def calc_top_k_eigenvalues_eigenvectors_symmetric(adj, k, smallest):
    """reference :764-814 -- takes k+1 and drops the first."""
    B, N, _ = adj.shape
    vals = torch.zeros(B, k + 1)
    vecs = torch.zeros(B, N, k + 1)
    all_vals = torch.zeros(B, N)
    all_vecs = torch.zeros(B, N, N)
    for i in range(B):
        A = adj[i]
        A = (A + A.t()) / 2
        Dm = torch.diag(torch.sum(A, dim=1))
        D_is = torch.diag(torch.pow(torch.diag(Dm), -0.5))
        Ls = torch.eye(N) - torch.matmul(torch.matmul(D_is, A), D_is)
        ev, evec = torch.linalg.eigh(Ls)
        tv, ti = torch.topk(ev, k + 1, largest=not smallest, sorted=True)
        vals[i] = tv
        vecs[i] = evec[:, ti]
        all_vals[i] = ev
        all_vecs[i] = evec
    return vals[:, 1:], vecs[:, :, 1:], all_vals, all_vecs


def sort_points_by_fiedler(points, fiedler):
    """reference :817-826 (the hard-coded 384 generalised to points.shape[-1])."""
    _, order = torch.sort(fiedler, dim=1)
    return torch.gather(points, 1, order.unsqueeze(-1).expand(-1, -1, points.shape[-1]))


def multilevel_travers(eigvecs, level):
    """reference :829-841."""
    means = eigvecs.mean(dim=1, keepdim=True)
    bits = (eigvecs >= means)[:, :, :level]
    pw = 2 ** torch.arange(level - 1, -1, -1)
    return torch.sum(bits * pw[None, None, :], dim=-1)


def spectral_orders(vecs):
    """(B,G,k) eigenvectors -> (B,k,G) int64 ascending argsort per eigenvector (:820)."""
    return torch.sort(vecs.transpose(1, 2), dim=2)[1]


def sast_assemble(tokens, pos, vecs, reverse=True):
    """reference :889-898 + :982-989: concatenate the k orderings, then append the flip."""
    k = vecs.shape[2]
    xs = [sort_points_by_fiedler(tokens, vecs[:, :, i]) for i in range(k)]
    ps = [sort_points_by_fiedler(pos, vecs[:, :, i]) for i in range(k)]
    x, p = torch.cat(xs, 1), torch.cat(ps, 1)
    if reverse:
        x = torch.cat((x, x.flip(1)), 1)
        p = torch.cat((p, p.flip(1)), 1)
    return x, p


def sast_index_map(orders, reverse=True):
    """(B,k,G) orders -> (B, k*G*(1+reverse)) gather indices equivalent to sast_assemble."""
    idx = orders.flatten(1)
    if reverse:
        idx = torch.cat((idx, idx.flip(1)), 1)
    return idx


def hlt_order_and_assemble(tokens, pos, center, vecs, k, rand=None):
    """reference :1059-1112 (method == "HLT", reverse == True), restated slice by slice -- including the
    reference's overlapping block writes: forward block i lands in output block i+1 (i >= 1) and is then
    overwritten by nothing / overwrites the previous reverse block, so the result is
    [fwd0, rev0, fwd1, fwd2, ..., fwd_{nd-1}, rev_{nd-1}, zeros...].  ``rand`` replaces torch.rand (:1062)."""
    integers = multilevel_travers(vecs, k).to(tokens.dtype)
    if rand is not None:
        integers = integers + rand
    order = torch.argsort(integers, dim=1, stable=True)
    st = torch.gather(tokens, 1, order.unsqueeze(-1).expand(-1, -1, tokens.shape[-1]))
    sp = torch.gather(pos, 1, order.unsqueeze(-1).expand(-1, -1, pos.shape[-1]))
    sc = torch.gather(center, 1, order.unsqueeze(-1).expand(-1, -1, 3))
    ng = 2 ** k
    nd = int(sp.shape[1] / ng)
    out_t = torch.zeros(sp.shape[0], sp.shape[1] * 2, st.shape[2])
    out_p = torch.zeros(sp.shape[0], sp.shape[1] * 2, sp.shape[2])
    out_c = torch.zeros(sp.shape[0], sp.shape[1] * 2, 3)
    for i in range(nd):
        src = slice(i * ng, (i + 1) * ng)
        dst = slice(i * ng, (i + 1) * ng) if i == 0 else slice((i + 1) * ng, (i + 2) * ng)
        out_t[:, dst], out_p[:, dst], out_c[:, dst] = st[:, src], sp[:, src], sc[:, src]
        rdst = slice((i + 1) * ng, (i + 2) * ng) if i == 0 else slice((i + 2) * ng, (i + 3) * ng)
        out_t[:, rdst], out_p[:, rdst], out_c[:, rdst] = st[:, src].flip(1), sp[:, src].flip(1), sc[:, src].flip(1)
    return out_t, out_p, out_c, order
