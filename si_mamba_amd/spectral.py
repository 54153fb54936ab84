"""Spectral token ordering on the MI355X kernels.

Free-function counterparts of the reference's ``PointMamba`` methods
(models/point_mamba.py), same argument names and return values:

  create_graph_from_centers                                :620-661
  create_graph_from_feature_space_gpu_weighted_adjacency   :664-715
  calc_top_k_eigenvalues_eigenvectors                      :717-761
  calc_top_k_eigenvalues_eigenvectors_symmetric            :764-814
  sort_points_by_fiedler                                   :817-826
  multilevel_travers                                       :829-841

plus the fused fast path ``spectral_order`` (centres -> eigenpairs + argsort orders in two
launches, no host synchronisation, replacing the reference's per-sample Python loop) and
``sast_gather`` (the token assembly of :889-898 / :982-989 as ONE gather per tensor).
``bind_to(cls)`` monkey-patches the methods onto a reference-style PointMamba class.
"""
from __future__ import annotations

import torch

from . import _lib


def _flags(symmetric=False, self_loop=False, binary=False, matrix_sym=False, smallest=False,
           sigma_mean=False):
    f = 0
    f |= _lib.SPEC_SYMMETRIC if symmetric else 0
    f |= _lib.SPEC_SELF_LOOP if self_loop else 0
    f |= _lib.SPEC_BINARY if binary else 0
    f |= _lib.SPEC_MATRIX_SYM if matrix_sym else 0
    f |= _lib.SPEC_SMALLEST if smallest else 0
    f |= _lib.SPEC_SIGMA_MEAN if sigma_mean else 0
    return f


def _knn_graph(points, k, alpha, flags):
    _lib.require_gpu(points, "create_graph")
    lib = _lib.load()
    pts = points.detach().float().contiguous()
    B, G, F = pts.shape
    adj = torch.empty(B, G, G, device=pts.device, dtype=torch.float32)
    ws = torch.empty(256, device=pts.device, dtype=torch.uint8)
    with torch.cuda.device(pts.device):
        rc = lib.simamba_knn_graph(_lib.ptr(pts), _lib.ptr(adj), _lib.ptr(ws), ws.numel(), B, G, F, int(k),
                                   float(alpha), flags, _lib.stream_ptr(pts.device))
    _lib.check(rc, "simamba_knn_graph")
    return adj


def create_graph_from_feature_space_gpu_weighted_adjacency(points, k=5, alpha=1, symmetric=False,
                                                           self_loop=False, binary=False):
    """(B,G,F) points -> (B,G,G) adjacency; reference :664-715."""
    return _knn_graph(points, k, alpha, _flags(symmetric, self_loop, binary))


def create_graph_from_centers(points, k=5, alpha=1, symmetric=False, self_loop=False, binary=False,
                              self_alpha=None):
    """reference :620-661.  The reference tests the *module attribute* alpha against 0 (:647) to pick
    the sigma = mean-distance weighting; pass it as ``self_alpha`` (defaults to ``alpha``)."""
    sigma_mean = (alpha if self_alpha is None else self_alpha) == 0
    return _knn_graph(points, k, alpha, _flags(symmetric, self_loop, binary, sigma_mean=sigma_mean))


def _eig(adj, k, smallest, matrix_sym, want_all=True, want_order=False):
    _lib.require_gpu(adj, "calc_top_k_eigenvalues_eigenvectors")
    lib = _lib.load()
    a = adj.detach().float().contiguous()
    B, G, _ = a.shape
    dev = a.device
    vals = torch.empty(B, k, device=dev, dtype=torch.float32)
    vecs = torch.empty(B, G, k, device=dev, dtype=torch.float32)
    order = torch.empty(B, k, G, device=dev, dtype=torch.int64) if want_order else None
    all_vals = torch.empty(B, G, device=dev, dtype=torch.float32) if want_all else None
    all_vecs = torch.empty(B, G, G, device=dev, dtype=torch.float32) if want_all else None
    with torch.cuda.device(dev):
        rc = lib.simamba_laplacian_topk(_lib.ptr(a), _lib.ptr(vals), _lib.ptr(vecs), _lib.ptr(order),
                                        _lib.ptr(all_vals), _lib.ptr(all_vecs), B, G, int(k),
                                        _flags(matrix_sym=matrix_sym, smallest=smallest),
                                        _lib.stream_ptr(dev))
    _lib.check(rc, "simamba_laplacian_topk")
    return vals, vecs, all_vals, all_vecs, order


def calc_top_k_eigenvalues_eigenvectors(adj_matrices, k, smallest):
    """reference :717-761 -> (vals (B,k), vecs (B,G,k), all_vals (B,G), all_vecs (B,G,G))."""
    return _eig(adj_matrices, k, smallest, matrix_sym=False)[:4]


def calc_top_k_eigenvalues_eigenvectors_symmetric(adj_matrices, k, smallest):
    """reference :764-814 (symmetric normalised Laplacian, first selected eigenpair dropped)."""
    return _eig(adj_matrices, k, smallest, matrix_sym=True)[:4]


def argsort_rows(vals):
    """(rows, n) fp32 -> (rows, n) int64 ascending argsort, ties by index (torch.sort at :820)."""
    _lib.require_gpu(vals, "argsort_rows")
    lib = _lib.load()
    v = vals.detach().float().contiguous()
    rows, n = v.shape
    idx = torch.empty(rows, n, device=v.device, dtype=torch.int64)
    with torch.cuda.device(v.device):
        rc = lib.simamba_argsort_rows(_lib.ptr(v), _lib.ptr(idx), rows, n, _lib.stream_ptr(v.device))
    _lib.check(rc, "simamba_argsort_rows")
    return idx


def sort_points_by_fiedler(points, fiedler_vector):
    """reference :817-826 (384 generalised to points.shape[-1]); differentiable w.r.t. points."""
    order = argsort_rows(fiedler_vector)
    return torch.gather(points, 1, order.unsqueeze(-1).expand(-1, -1, points.shape[-1]))


def multilevel_travers(eigen_vectors, level):
    """reference :829-841 (pure index arithmetic; stays in torch)."""
    means = eigen_vectors.mean(dim=1, keepdim=True)
    binaries = (eigen_vectors >= means)[:, :, :level]
    powers = 2 ** torch.arange(start=level - 1, end=-1, step=-1, device=eigen_vectors.device)
    return torch.sum(binaries * powers[None, None, :], dim=-1, keepdim=True).squeeze()


def spectral_order(center, knn_graph, alpha, k_top_eigenvectors, smallest=True, symmetric=False,
                   self_loop=False, binary=False, matrix="laplacian"):
    """Fused SAST ordering: centres (B,G,3) -> (vals (B,k), vecs (B,G,k), order (B,k,G) int64).

    Equivalent to reference :872 + :884 + the k argsorts of :889-890, without materialising
    anything on the host.
    """
    _lib.require_gpu(center, "spectral_order")
    lib = _lib.load()
    c = center.detach().float().contiguous()
    B, G, F = c.shape
    if F != 3:
        raise ValueError("spectral_order expects (B, G, 3) centres")
    dev = c.device
    k = int(k_top_eigenvectors)
    vals = torch.empty(B, k, device=dev, dtype=torch.float32)
    vecs = torch.empty(B, G, k, device=dev, dtype=torch.float32)
    order = torch.empty(B, k, G, device=dev, dtype=torch.int64)
    nbytes = lib.simamba_spectral_workspace_bytes(B, G)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    flags = _flags(symmetric, self_loop, binary, matrix_sym=(matrix != "laplacian"), smallest=smallest)
    with torch.cuda.device(dev), _lib.timed("spectral_topk", dev):
        rc = lib.simamba_spectral_topk(_lib.ptr(c), _lib.ptr(vals), _lib.ptr(vecs), _lib.ptr(order),
                                       _lib.ptr(ws), nbytes, B, G, int(knn_graph), float(alpha), k, flags,
                                       _lib.stream_ptr(dev))
    _lib.check(rc, "simamba_spectral_topk")
    return vals, vecs, order


def sast_index_map(order, reverse=True):
    """(B,k,G) orders -> (B, k*G*(1+reverse)) token indices: k orderings, then their flip (:982-989)."""
    idx = order.flatten(1)
    if reverse:
        idx = torch.cat((idx, idx.flip(1)), 1)
    return idx


def sast_gather(tokens, pos, order, reverse=True):
    """Token assembly of reference :889-898 + :982-989 as one gather per tensor."""
    idx = sast_index_map(order, reverse)
    ex = idx.unsqueeze(-1)
    return (torch.gather(tokens, 1, ex.expand(-1, -1, tokens.shape[-1])),
            torch.gather(pos, 1, ex.expand(-1, -1, pos.shape[-1])))


_hlt_maps = {}


def hlt_index_map(G, k, device=None):
    """Source position (into the code-sorted sequence of G patches) of each of the 2G output slots of the
    reference's HLT assembly (:1075-1112, reverse == True), -1 where the reference leaves zeros.

    The reference writes block i (2^k patches) forward into output block i+1 (block 0 for i == 0) and flipped
    into block i+2 (block 1 for i == 0); later writes overwrite earlier ones.  Replaying those assignments on
    an index vector gives the same result as its slice assignments, in one gather.
    """
    key = (G, k, None if device is None else str(device))
    if key in _hlt_maps:                      # built once per shape and device: no host-to-device copy per call
        return _hlt_maps[key]                 # (and none inside a hipGraph capture)
    ng = 2 ** k
    nd = G // ng
    idx = torch.full((2 * G,), -1, dtype=torch.int64)
    for i in range(nd):
        src = torch.arange(i * ng, (i + 1) * ng)
        d0 = i * ng if i == 0 else (i + 1) * ng
        r0 = (i + 1) * ng if i == 0 else (i + 2) * ng
        if d0 + ng <= 2 * G:
            idx[d0:d0 + ng] = src
        if r0 + ng <= 2 * G:
            idx[r0:r0 + ng] = src.flip(0)
    idx = idx if device is None else idx.to(device)
    _hlt_maps[key] = idx
    return idx


def hlt_assemble(tokens, pos, center, top_k_eigenvectors, k, rand=None):
    """HLT token order of reference :1059-1112: bit-code traversal (multilevel_travers) + random tie-break
    (``rand``: (B, G) in [0,1), the reference's torch.rand at :1062; None = no tie-break) -> argsort -> the
    block assembly above.  Returns (tokens (B,2G,C), pos (B,2G,C), center (B,2G,3), order (B,G))."""
    codes = multilevel_travers(top_k_eigenvectors, k).to(torch.float32)
    if codes.dim() == 1:
        codes = codes.unsqueeze(0)
    if rand is not None:
        codes = codes + rand
    order = argsort_rows(codes)
    G = order.shape[1]
    slot = hlt_index_map(G, k, order.device)
    valid = (slot >= 0)
    src = torch.gather(order, 1, slot.clamp_min(0).unsqueeze(0).expand(order.shape[0], -1))     # (B, 2G)

    def pick(x):
        out = torch.gather(x, 1, src.unsqueeze(-1).expand(-1, -1, x.shape[-1]))
        return out * valid.to(x.dtype)[None, :, None]

    return pick(tokens), pick(pos), pick(center), order


_side_streams = {}


def run_on_side_stream(fn, *inputs):
    """Start ``fn()`` on a side HIP stream (one per device, created on first use) and return a ``join()`` that makes
    the current stream wait for it and hands back its result.  The eigen-ordering kernels occupy B of the 256 CUs
    and depend only on the patch centres, so callers launch them underneath the (GEMM-bound) patch encoder.
    ``inputs``: the tensors ``fn`` reads (kept alive for the side stream).  CPU tensors: runs inline."""
    if not inputs or not inputs[0].is_cuda:
        res = fn()
        return lambda: res
    dev = inputs[0].device
    main = torch.cuda.current_stream(dev)
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        res = fn()
    for t in inputs:
        t.record_stream(side)

    def join():
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(side)
        for r in (res if isinstance(res, (tuple, list)) else (res,)):
            if torch.is_tensor(r):
                r.record_stream(cur)
        return res
    return join


def bind_to(cls):
    """Monkey-patch the spectral methods of a reference-style PointMamba class with these kernels."""
    def _m(fn):
        return lambda self, *a, **kw: fn(*a, **kw)

    def _centers(self, points, k=5, alpha=1, symmetric=False, self_loop=False, binary=False):
        return create_graph_from_centers(points, k, alpha, symmetric, self_loop, binary,
                                         self_alpha=getattr(self, "alpha", alpha))

    cls.create_graph_from_centers = _centers
    cls.create_graph_from_feature_space_gpu_weighted_adjacency = _m(
        create_graph_from_feature_space_gpu_weighted_adjacency)
    cls.calc_top_k_eigenvalues_eigenvectors = _m(calc_top_k_eigenvalues_eigenvectors)
    cls.calc_top_k_eigenvalues_eigenvectors_symmetric = _m(calc_top_k_eigenvalues_eigenvectors_symmetric)
    cls.sort_points_by_fiedler = _m(sort_points_by_fiedler)
    cls.multilevel_travers = _m(multilevel_travers)
    return cls
