"""Generate tests/golden/*.npz and the parameter-table fixture.  Run in the build container:

    python -m oracle.gen_golden            # from the repo root

What pins what
  * scan_*.npz / conv_*.npz / mamba_block_cfg1.npz / spectral_*.npz / interp_seg.npz / chamfer_mae.npz are
    produced by THIS repo's
    CPU restatement (oracle/) with stock CPU torch; they are regression pins of the oracle and
    the inputs/expected outputs of the GPU parity tests.  The reference cannot be imported here
    (its module-level imports need the absent wheels mamba_ssm, pytorch3d, timm, easydict), so
    these are NOT reference outputs: parity stays "unpinned" for them (Oracle).
  * param_table_finetune_hardest.json is transcribed from the reference's own training log
    (logs/finetuned_hardest.log:100-426; read as text, needs /root/reference) and pins the
    state-dict contract.
"""
from __future__ import annotations

import json
import math
import os
import re
import sys

import numpy as np
import torch

from oracle import mae_ref, scan_ref, seg_ref, spectral_ref
from si_mamba_amd.synthetic import scan_inputs, unit_ball_centers  # noqa: F401  (re-exported for the tests)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF_LOG = "/root/reference/logs/finetuned_hardest.log"


def scan_case(name, **kw):
    inp = scan_inputs(**kw)
    leaf = {k: (v.clone().requires_grad_(True) if v is not None and k != "dout" else v) for k, v in inp.items()}
    out, last = scan_ref.selective_scan_ref(leaf["u"], leaf["delta"], leaf["A"], leaf["B"], leaf["C"], leaf["D"],
                                            leaf["z"], leaf["delta_bias"], delta_softplus=True,
                                            return_last_state=True)
    out.backward(inp["dout"])
    rec = {k: v.numpy() for k, v in inp.items() if v is not None}
    rec["out"] = out.detach().numpy()
    rec["last_state"] = last.detach().numpy()
    for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias"):
        if leaf[k] is not None:
            rec["grad_" + k] = leaf[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def conv_case(name, batch, dim, L, W, act, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, dim, L, generator=g, requires_grad=True)
    w = (torch.randn(dim, W, generator=g) * 0.5).requires_grad_(True)
    b = torch.randn(dim, generator=g).requires_grad_(True) if bias else None
    dout = torch.randn(batch, dim, L, generator=g)
    out = scan_ref.causal_conv1d_ref(x, w, b, act)
    out.backward(dout)
    rec = dict(x=x.detach().numpy(), w=w.detach().numpy(), dout=dout.numpy(), out=out.detach().numpy(),
               grad_x=x.grad.numpy(), grad_w=w.grad.numpy(), silu=np.array(int(act is not None)))
    if b is not None:
        rec["bias"] = b.detach().numpy()
        rec["grad_bias"] = b.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def mamba_block_case():
    """BASELINE config 1: single mixer, B=2, L=64, d_model=128, d_state=16."""
    torch.manual_seed(0)
    m = scan_ref.MambaRef(128, layer_idx=0)
    g = torch.Generator().manual_seed(1)
    h = torch.randn(2, 64, 128, generator=g, requires_grad=True)
    dout = torch.randn(2, 64, 128, generator=g)
    out = m(h)
    out.backward(dout)
    rec = {"hidden": h.detach().numpy(), "dout": dout.numpy(), "out": out.detach().numpy(),
           "grad_hidden": h.grad.numpy()}
    for k, v in m.state_dict().items():
        rec["param." + k] = v.numpy()
    for k, p in m.named_parameters():
        rec["grad." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "mamba_block_cfg1.npz"), **rec)


SPECTRAL_COMBOS = [
    # the flag sets the reference's configs use
    dict(tag="hardest", knn=20, alpha=10.0, symmetric=True, self_loop=False, binary=True),      # scan_hardest / pretrain
    dict(tag="modelnet", knn=20, alpha=100.0, symmetric=True, self_loop=False, binary=True),
    dict(tag="objbg", knn=20, alpha=10.0, symmetric=True, self_loop=True, binary=False),
    dict(tag="partseg", knn=10, alpha=10.0, symmetric=True, self_loop=True, binary=False),
    dict(tag="asym", knn=8, alpha=4.0, symmetric=False, self_loop=False, binary=False),
]


def surface_centers(B, G, seed, npoints=1024):
    """Patch centres as the pipeline makes them: farthest-point sampling (oracle/fps_ref.py) of clouds on thin surfaces."""
    from oracle import fps_ref
    from si_mamba_amd.synthetic import surface_clouds
    pts = surface_clouds(B, npoints, seed)
    idx = fps_ref.sample_farthest_points(pts, G)
    idx = idx[1] if isinstance(idx, tuple) else idx
    return torch.gather(pts, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()


def spectral_case(name, B, G, seed, k=4, centers=None):
    centers = unit_ball_centers(B, G, seed) if centers is None else centers
    rec = {"centers": centers.numpy()}
    for cb in SPECTRAL_COMBOS:
        adj = spectral_ref.create_graph_from_feature_space(centers, cb["knn"], cb["alpha"], cb["symmetric"],
                                                           cb["self_loop"], cb["binary"])
        vals, vecs, all_vals, _ = spectral_ref.calc_top_k_eigenvalues_eigenvectors(adj, k, True)
        order = spectral_ref.spectral_orders(vecs)
        svec = torch.sort(vecs.transpose(1, 2), dim=2)[0]
        min_gap = (svec[:, :, 1:] - svec[:, :, :-1]).min(dim=2)[0]           # (B,k)
        t = cb["tag"]
        rec[f"{t}.adj"] = adj.numpy()
        rec[f"{t}.vals"] = vals.numpy()
        rec[f"{t}.vecs"] = vecs.numpy()
        rec[f"{t}.all_vals"] = all_vals.numpy()
        rec[f"{t}.order"] = order.numpy()
        rec[f"{t}.min_gap"] = min_gap.numpy()
    # symmetric-normalised variant and "largest" selection on the first combo
    cb = SPECTRAL_COMBOS[0]
    adj = spectral_ref.create_graph_from_feature_space(centers, cb["knn"], cb["alpha"], cb["symmetric"],
                                                       cb["self_loop"], cb["binary"])
    v, e, _, _ = spectral_ref.calc_top_k_eigenvalues_eigenvectors_symmetric(adj, k, True)
    rec["hardest.sym.vals"], rec["hardest.sym.vecs"] = v.numpy(), e.numpy()
    v, e, _, _ = spectral_ref.calc_top_k_eigenvalues_eigenvectors(adj, k, False)
    rec["hardest.largest.vals"], rec["hardest.largest.vecs"] = v.numpy(), e.numpy()
    # alpha == 0 branch of create_graph_from_centers (sigma = mean distance over the batch)
    adj0 = spectral_ref.create_graph_from_centers(centers, 10, 0.0, True, True, False)
    rec["sigma_mean.adj"] = adj0.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def interp_case(name, B, N, S, C, seed):
    """3-NN inverse-distance interpolation (part_segmentation/models/pointnet2_utils.py:285-297)."""
    pts = unit_ball_centers(B, N, seed)
    g = torch.Generator().manual_seed(seed)
    centres = pts[:, torch.randperm(N, generator=g)[:S]].clone()
    feats = torch.randn(B, S, C, generator=g)
    out, idx, w = seg_ref.three_nn_interpolate(pts, centres, feats)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), xyz1=pts.numpy(), xyz2=centres.numpy(), feats=feats.numpy(),
                        out=out.numpy(), idx=idx.numpy().astype(np.int32), weight=w.numpy())


def chamfer_case(name, pairs, n, m, seed):
    """Chamfer-L2 of the MAE loss (pytorch3d semantics, models/point_mamba.py:3203)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(pairs, n, 3, generator=g, requires_grad=True)
    y = torch.randn(pairs, m, 3, generator=g)
    w = torch.rand(pairs, generator=g)
    d = mae_ref.chamfer_distance(x, y)
    (d * w).sum().backward()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), pred=x.detach().numpy(), gt=y.numpy(), wsum=w.numpy(),
                        dist=d.detach().numpy(), grad_pred=x.grad.numpy())


def param_table():
    if not os.path.exists(REF_LOG):
        print("reference log not present; keeping the committed param table", file=sys.stderr)
        return
    pat = re.compile(r"\|module\.(\S+)\s*\|torch\.(\w+)\s*\|\(([^)]*)\)\s*\|(\d+)\s*\|")
    rows = []
    with open(REF_LOG, "r", errors="replace") as fh:
        for ln, line in enumerate(fh, 1):
            if ln > 430:
                break
            m = pat.search(line)
            if m:
                shape = [int(s) for s in m.group(3).replace(" ", "").split(",") if s]
                rows.append({"name": m.group(1), "dtype": m.group(2), "shape": shape, "numel": int(m.group(4))})
    total = sum(r["numel"] for r in rows)
    with open(os.path.join(OUT, "param_table_finetune_hardest.json"), "w") as fh:
        json.dump({"source": "reference logs/finetuned_hardest.log:100-426", "total_numel": total,
                   "params": rows}, fh, indent=0)
    print("param table:", len(rows), "tensors,", total, "parameters")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    scan_case("scan_cfg1", batch=2, dim=64, L=64, N=16, seed=0)
    scan_case("scan_l128", batch=2, dim=48, L=128, N=16, seed=1)
    scan_case("scan_multichunk_ragged", batch=1, dim=32, L=300, N=16, seed=2)
    scan_case("scan_odd", batch=2, dim=20, L=37, N=5, seed=3, with_z=False, with_D=False, with_bias=False)
    conv_case("conv_cfg1", 2, 64, 64, 4, "silu", 0)
    conv_case("conv_odd", 1, 20, 37, 3, None, 1, bias=False)
    mamba_block_case()
    spectral_case("spectral_g64", 4, 64, 0)
    spectral_case("spectral_g128", 2, 128, 1)
    spectral_case("spectral_g128_surface", 4, 128, 7, centers=surface_centers(4, 128, 7))
    interp_case("interp_seg", 2, 256, 32, 48, 4)
    chamfer_case("chamfer_mae", 64, 32, 32, 5)
    param_table()


if __name__ == "__main__":
    main()
