"""Pin the oracle to the reference's OWN code: run the reference's function bodies here, save what they return.

    python -m oracle.pin_from_reference          # from the repo root, in the build container (needs /root/reference)

TEST INFRASTRUCTURE.  Writes tests/golden/ref_*.npz; nothing of the reference's source is copied into the repository
-- the functions are read from /root/reference at run time, executed, and only their inputs and outputs are kept.

Why not ``import models.point_mamba``: its module-level imports need timm, pytorch3d, mamba_ssm and easydict, none of
which exist in this image (an ordinary ImportError, nothing was denied).  The functions on the hot path need none of
them.  So this script
  * parses the two reference files with ``ast`` and takes the definitions it needs BY NAME -- methods of
    ``PointMamba`` (models/point_mamba.py:620-841: create_graph_from_centers,
    create_graph_from_feature_space_gpu_weighted_adjacency, calc_top_k_eigenvalues_eigenvectors[_symmetric],
    sort_points_by_fiedler, multilevel_travers), the SAST token assembly and the HLT ordering / block assembly out of
    ``PointMamba.forward`` (the statements at :889-898, :982-989 and :1059-1112), ``_init_weights``, ``create_block``,
    ``MixerModel`` (:115-272) and ``Block`` (models/block.py:17-76);
  * compiles them UNMODIFIED and calls them with stock CPU torch under a ``TorchFunctionMode`` that maps the hard-coded
    ``device='cuda'`` / ``.cuda()`` / ``.to('cuda')`` to the CPU (this container has no GPU);
  * gives ``create_block`` the oracle's ``MambaRef`` as its ``Mamba`` (the real one lives in the absent mamba-ssm wheel:
    that half stays "parity unpinned") and ``Block`` a pass-through ``DropPath`` (never executed at drop_path = 0).

What the fixtures pin: adjacency, eigenpairs (same LAPACK as tests/test_oracle_spectral.py runs), orderings, the SAST
index map, the HLT order and slot map, the Add -> LayerNorm -> mixer data flow of Block / MixerModel and the
initialisation contract of _init_weights.  tests/test_oracle_pinned.py holds the oracle to them on the CPU;
tests/test_gpu_pinned.py holds the HIP path to the same files on the GPU box (where /root/reference does not exist).
"""
from __future__ import annotations

import ast
import math
import os
import sys
from functools import partial
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
REF_MODEL = os.path.join(REF, "models", "point_mamba.py")
REF_BLOCK = os.path.join(REF, "models", "block.py")

SPECTRAL_METHODS = ("create_graph_from_centers", "create_graph_from_feature_space_gpu_weighted_adjacency",
                    "calc_top_k_eigenvalues_eigenvectors", "calc_top_k_eigenvalues_eigenvectors_symmetric",
                    "sort_points_by_fiedler", "multilevel_travers")


def reference_present():
    return os.path.exists(REF_MODEL) and os.path.exists(REF_BLOCK)


class CudaToCpu(TorchFunctionMode):
    """device='cuda' -> cpu, tensor.cuda() / tensor.to('cuda') -> the tensor itself."""

    @staticmethod
    def _is_cuda(d):
        return (isinstance(d, str) and d.startswith("cuda")) or (isinstance(d, torch.device) and d.type == "cuda")

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.to and len(args) >= 2 and self._is_cuda(args[1]):
            return args[0] if len(args) == 2 and not kwargs else func(args[0], "cpu", *args[2:], **kwargs)
        if self._is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def _parse(path):
    with open(path, "r") as fh:
        return ast.parse(fh.read(), filename=path)


def _find(body, kind, name):
    for node in body:
        if isinstance(node, kind) and node.name == name:
            return node
    raise LookupError(f"{name} not found in the reference")


def _exec_defs(nodes, scope, filename):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, filename, "exec"), scope)
    return scope


def _stmts_between(fn, first, last):
    """Statements of a function body (searched recursively through if / elif / for bodies) that start in [first, last]
    at the nesting depth of the first one found."""
    def walk(body):
        hit = [n for n in body if first <= n.lineno <= last]
        if hit and hit[0].lineno == first:
            return hit
        for n in body:
            for field in ("body", "orelse"):
                sub = getattr(n, field, None)
                if isinstance(sub, list) and sub and isinstance(sub[0], ast.stmt):
                    r = walk(sub)
                    if r:
                        return r
        return None
    r = walk(fn.body)
    if not r:
        raise LookupError(f"no statement starts at line {first} of {fn.name}")
    return r


class _DropPath(nn.Module):
    """Stand-in for timm.models.layers.DropPath at drop_prob = 0 (the reference builds nn.Identity then; this class
    only has to exist so that the module-level name resolves)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        assert drop_prob == 0.0, "the pinned fixtures are generated at drop_path = 0"

    def forward(self, x):
        return x


def load_reference():
    """-> (methods: name -> function(self, ...), sast: callable, hlt: callable, scope with Block / MixerModel / ...)."""
    from oracle.scan_ref import MambaRef

    def Mamba(d_model, layer_idx=None, device=None, dtype=None, **kw):
        """create_block passes upstream's factory keywords (models/point_mamba.py:161-162); the oracle mixer is CPU fp32."""
        assert device is None and dtype is None
        return MambaRef(d_model, layer_idx=layer_idx, **kw)

    tree = _parse(REF_MODEL)
    cls = _find(tree.body, ast.ClassDef, "PointMamba")
    scope = {"torch": torch, "nn": nn, "F": F, "np": np, "math": math, "partial": partial, "Tensor": Tensor,
             "Optional": Optional, "Mamba": Mamba, "DropPath": _DropPath, "RMSNorm": None, "layer_norm_fn": None,
             "rms_norm_fn": None}
    _exec_defs([_find(cls.body, ast.FunctionDef, m) for m in SPECTRAL_METHODS], scope, REF_MODEL)
    methods = {m: scope[m] for m in SPECTRAL_METHODS}

    # the token assembly of the SAST branch and the HLT branch, cut out of PointMamba.forward as they stand
    fwd = _find(cls.body, ast.FunctionDef, "forward")
    sast_loop = _stmts_between(fwd, 889, 898)
    sast_rev = _stmts_between(fwd, 982, 989)
    assert isinstance(sast_loop[0], ast.For) and "sort_points_by_fiedler" in ast.unparse(sast_loop[0]), \
        "reference layout changed: SAST ordering loop not at models/point_mamba.py:889"
    assert isinstance(sast_rev[0], ast.If) and "reverse" in ast.unparse(sast_rev[0].test), \
        "reference layout changed: SAST reverse block not at models/point_mamba.py:982"
    hlt = _stmts_between(fwd, 1059, 1112)
    assert "multilevel_travers" in ast.unparse(hlt[0]), "reference layout changed: HLT block not at :1059"

    def make(name, stmts, argnames, result):
        fn = ast.FunctionDef(name=name, args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in argnames],
                                                           kwonlyargs=[], kw_defaults=[], defaults=[]),
                             body=list(stmts) + [ast.Return(value=ast.Tuple(
                                 elts=[ast.Name(id=r, ctx=ast.Load()) for r in result], ctx=ast.Load()))],
                             decorator_list=[])
        _exec_defs([fn], scope, REF_MODEL)
        return scope[name]

    sast = make("_ref_sast", [sast_loop[0], sast_rev[0]], ["self", "group_input_tokens", "pos", "top_k_eigenvectors"],
                ["group_input_tokens", "pos"])
    hlt_fn = make("_ref_hlt", hlt, ["self", "group_input_tokens", "pos", "center", "top_k_eigenvectors"],
                  ["group_input_tokens", "pos", "sorted_center", "integers_arg_sort", "integers_before_random"])

    # Block (models/block.py) and the stack builders (models/point_mamba.py)
    btree = _parse(REF_BLOCK)
    _exec_defs([_find(btree.body, ast.ClassDef, "Block")], scope, REF_BLOCK)
    _exec_defs([_find(tree.body, ast.FunctionDef, "_init_weights"), _find(tree.body, ast.FunctionDef, "create_block"),
                _find(tree.body, ast.ClassDef, "MixerModel")], scope, REF_MODEL)
    return methods, sast, hlt_fn, scope


class RefSelf:
    """The attributes of PointMamba the extracted code reads, and its methods bound."""

    def __init__(self, methods, **attrs):
        self.k_top_eigenvectors, self.reverse, self.alpha = 4, True, 10.0
        for k, v in attrs.items():
            setattr(self, k, v)
        for name, fn in methods.items():
            setattr(self, name, fn.__get__(self))


def index_tokens(B, G, base):
    """(B, G, 384) tokens whose every channel is base + the token's index: what a gather leaves in channel 0 IS the
    index map it applied (sort_points_by_fiedler hard-codes 384 channels, models/point_mamba.py:822)."""
    return (base + torch.arange(G, dtype=torch.float32))[None, :, None].expand(B, G, 384).contiguous()


def spectral_fixture(name, methods, sast, hlt_fn):
    """Reference outputs on the centres of an existing fixture tests/golden/<name>.npz -> ref_<name>.npz."""
    from oracle.gen_golden import SPECTRAL_COMBOS
    centers = torch.from_numpy(np.load(os.path.join(OUT, name + ".npz"))["centers"])
    B, G, _ = centers.shape
    k = 4
    # eigh's last bits depend on the LAPACK build and its thread count: recorded so that a consumer can reproduce them
    rec = {"centers": centers.numpy(), "lapack_threads": np.array(torch.get_num_threads()),
           "torch_version": np.array(torch.__version__)}
    tok, pos = index_tokens(B, G, 0.0), index_tokens(B, G, 1000.0)
    with CudaToCpu():
        for cb in SPECTRAL_COMBOS:
            me = RefSelf(methods, alpha=cb["alpha"])
            adj = me.create_graph_from_feature_space_gpu_weighted_adjacency(
                centers, cb["knn"], cb["alpha"], cb["symmetric"], cb["self_loop"], cb["binary"])
            vals, vecs, all_vals, all_vecs = me.calc_top_k_eigenvalues_eigenvectors(adj, k, True)
            order = torch.stack([me.sort_points_by_fiedler(tok, vecs[:, :, i])[:, :, 0] for i in range(k)], 1)
            x, p = sast(me, tok, pos, vecs)
            assert torch.equal(p, x + 1000.0)                  # tokens and pos go through the same map
            t = cb["tag"]
            rec[f"{t}.adj"] = adj.numpy()
            rec[f"{t}.vals"], rec[f"{t}.vecs"] = vals.numpy(), vecs.numpy()
            rec[f"{t}.all_vals"] = all_vals.numpy()
            rec[f"{t}.order"] = order.long().numpy()           # (B, k, G)
            rec[f"{t}.sast_index"] = x[:, :, 0].long().numpy()  # (B, 2 k G)
        cb = SPECTRAL_COMBOS[0]
        me = RefSelf(methods, alpha=cb["alpha"])
        adj = me.create_graph_from_feature_space_gpu_weighted_adjacency(
            centers, cb["knn"], cb["alpha"], cb["symmetric"], cb["self_loop"], cb["binary"])
        v, e, _, _ = me.calc_top_k_eigenvalues_eigenvectors_symmetric(adj, k, True)
        rec["hardest.sym.vals"], rec["hardest.sym.vecs"] = v.numpy(), e.numpy()
        v, e, _, _ = me.calc_top_k_eigenvalues_eigenvectors(adj, k, False)
        rec["hardest.largest.vals"], rec["hardest.largest.vecs"] = v.numpy(), e.numpy()
        # create_graph_from_centers: the alpha == 0 branch reads self.alpha (sigma = mean distance of the batch), and
        # the weighted form the segmentation config uses
        me0 = RefSelf(methods, alpha=0)
        rec["sigma_mean.adj"] = me0.create_graph_from_centers(centers, 10, 0.0, True, True, False).numpy()
        mew = RefSelf(methods, alpha=10.0)
        adjc = mew.create_graph_from_centers(centers, 10, 10.0, True, True, False)
        rec["centers_graph.adj"] = adjc.numpy()
        # HLT (models/point_mamba.py:1056-1112) on that graph, k = 3 levels, with the reference's own torch.rand
        # tie-break drawn from a seeded generator: the consumer redraws it with the same seed
        meh = RefSelf(methods, alpha=10.0, k_top_eigenvectors=3)
        _, hv, _, _ = meh.calc_top_k_eigenvalues_eigenvectors(adjc, 3, True)
        torch.manual_seed(1234)
        ht, hp, hc, horder, hcodes = hlt_fn(meh, index_tokens(B, G, 1.0), index_tokens(B, G, 1001.0), centers, hv)
        rec["hlt.vecs"] = hv.numpy()
        rec["hlt.codes"] = hcodes.long().numpy()
        rec["hlt.order"] = horder.long().numpy()
        rec["hlt.tokens_index"] = ht[:, :, 0].numpy()          # (B, 2 G): 1 + token index per slot, 0 where nothing is written
        rec["hlt.pos_index"] = hp[:, :, 0].numpy()
        rec["hlt.center"] = hc.numpy()
        rec["hlt.rand_seed"] = np.array(1234)
    np.savez_compressed(os.path.join(OUT, "ref_" + name + ".npz"), **rec)
    return rec


def stack_fixture(scope):
    """The reference's own create_block / Block / MixerModel / _init_weights around the oracle mixer -> ref_stack.npz."""
    MixerModel, create_block = scope["MixerModel"], scope["create_block"]
    d, n_layer, B, L = 64, 3, 2, 24
    torch.manual_seed(7)
    with CudaToCpu():
        model = MixerModel(d_model=d, n_layer=n_layer, rms_norm=False, drop_path=0.0)
        g = torch.Generator().manual_seed(8)
        x = torch.randn(B, L, d, generator=g, requires_grad=True)
        pos = torch.randn(B, L, d, generator=g, requires_grad=True)
        dout = torch.randn(B, L, d, generator=g)
        out = model(x, pos)
        out.backward(dout)
        # one Block on its own, both call forms (models/block.py:56-58)
        blk = create_block(d, layer_idx=0)
        blk.load_state_dict(model.layers[0].state_dict())
        h = torch.randn(B, L, d, generator=g)
        r = torch.randn(B, L, d, generator=g)
        h1, r1 = blk(h, None)
        h2, r2 = blk(h, r)
    rec = {"x": x.detach().numpy(), "pos": pos.detach().numpy(), "dout": dout.numpy(), "out": out.detach().numpy(),
           "grad_x": x.grad.numpy(), "grad_pos": pos.grad.numpy(),
           "block.h": h.numpy(), "block.r": r.numpy(), "block.first.h": h1.detach().numpy(),
           "block.first.r": r1.detach().numpy(), "block.next.h": h2.detach().numpy(),
           "block.next.r": r2.detach().numpy(), "dims": np.array([d, n_layer, B, L])}
    names = []
    for k, v in model.state_dict().items():
        rec["param." + k] = v.numpy()
        names.append(k)
    for k, p_ in model.named_parameters():
        rec["grad." + k] = p_.grad.numpy()
    rec["param_names"] = np.array(names)
    # what _init_weights leaves behind (models/point_mamba.py:122-144): Linear biases zero unless _no_reinit,
    # out_proj.weight re-drawn kaiming_uniform(a = sqrt 5) / sqrt(n_layer)
    bound = 1.0 / math.sqrt(2 * d) / math.sqrt(n_layer)       # kaiming_uniform(a = sqrt 5): U(+-1/sqrt(fan_in))
    rec["init.out_proj_bound"] = np.array(bound)
    rec["init.out_proj_absmax"] = np.array([model.layers[i].mixer.out_proj.weight.abs().max().item()
                                            for i in range(n_layer)])
    np.savez_compressed(os.path.join(OUT, "ref_stack.npz"), **rec)
    return rec


def main():
    if not reference_present():
        print("reference not present: keeping the committed ref_*.npz", file=sys.stderr)
        return 1
    torch.set_num_threads(4)
    methods, sast, hlt_fn, scope = load_reference()
    for name in ("spectral_g64", "spectral_g128", "spectral_g128_surface"):
        rec = spectral_fixture(name, methods, sast, hlt_fn)
        print(f"ref_{name}.npz: {len(rec)} arrays")
    rec = stack_fixture(scope)
    print(f"ref_stack.npz: {len(rec)} arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
