"""CPU restatement of the reference's part-segmentation head.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  part_segmentation/models/pointnet2_utils.py  square_distance :19-38, index_points :41-57,
                                               PointNetFeaturePropagation.forward :277-311
  part_segmentation/models/pt_mamba.py         MixerModelForSegmentation.forward :390-416,
                                               get_model.forward :631-787 (everything after the token ordering)
in the reference's own channels-first (B, C, N) layout and op sequence, on plain torch ops.  The reference holds
no fixtures for these functions: parity unpinned, but every call below is the one the reference itself makes.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def square_distance(src, dst):
    """:19-38 -- the expanded form, in the reference's order of operations."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


def index_points(points, idx):
    """:41-57."""
    B = points.shape[0]
    view_shape = list(idx.shape)
    view_shape[1:] = [1] * (len(view_shape) - 1)
    repeat_shape = list(idx.shape)
    repeat_shape[0] = 1
    batch_indices = torch.arange(B, dtype=torch.long).view(view_shape).repeat(repeat_shape)
    return points[batch_indices, idx, :]


def three_nn_interpolate(xyz1, xyz2, points2):
    """:285-297 -- xyz1 (B,N,3), xyz2 (B,S,3), points2 (B,S,D) -> (B,N,D), idx (B,N,3), weight (B,N,3).
    ``stable=True``: the reference's sort leaves ties (duplicated centres) unspecified; lowest index is the
    convention of the HIP kernel."""
    B, N, _ = xyz1.shape
    dists = square_distance(xyz1, xyz2)
    dists, idx = dists.sort(dim=-1, stable=True)
    dists, idx = dists[:, :, :3], idx[:, :, :3]
    dist_recip = 1.0 / (dists + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    out = torch.sum(index_points(points2, idx) * weight.view(B, N, 3, 1), dim=2)
    return out, idx, weight


def _autocast_ops(io_dtype):
    """(conv, bn) as they behave under ``torch.autocast`` on the reference's CUDA path when ``io_dtype`` is given:
    conv1d is on autocast's low-precision list (operands cast to ``io_dtype``, fp32 accumulation, ``io_dtype``
    result); batch_norm is on neither list, so it runs in its input's dtype (fp32 arithmetic inside, result rounded
    to the input dtype).  Values are carried as fp32 between the roundings.  ``io_dtype=None``: plain fp32 modules."""
    if io_dtype is None:
        return (lambda c, t: c(t)), (lambda b, t: b(t))
    r = lambda t: t.to(io_dtype).float()
    conv = lambda c, t: r(F.conv1d(r(t), r(c.weight), None if c.bias is None else r(c.bias)))
    bn = lambda b, t: r(b(t))
    return conv, bn


def feature_propagation(fp, xyz1, xyz2, points1, points2, io_dtype=None):
    """PointNetFeaturePropagation.forward :277-311 with the module ``fp``'s parameters (mlp_convs / mlp_bns);
    channels-first arguments as in the reference: xyz1 (B,3,N), xyz2 (B,3,S), points1 (B,D,N), points2 (B,D,S)."""
    conv_op, bn_op = _autocast_ops(io_dtype)
    xyz1 = xyz1.permute(0, 2, 1)
    xyz2 = xyz2.permute(0, 2, 1)
    points2 = points2.permute(0, 2, 1)
    B, N, C = xyz1.shape
    _, S, _ = xyz2.shape
    if S == 1:
        interpolated = points2.repeat(1, N, 1)
    else:
        interpolated, _, _ = three_nn_interpolate(xyz1, xyz2, points2)
    if points1 is not None:
        new_points = torch.cat([points1.permute(0, 2, 1), interpolated], dim=-1)
    else:
        new_points = interpolated
    new_points = new_points.permute(0, 2, 1)
    for conv, bn in zip(fp.mlp_convs, fp.mlp_bns):
        new_points = F.relu(bn_op(bn, conv_op(conv, new_points)))
    return new_points


def mixer_taps(blocks, mixers, x, pos):
    """MixerModelForSegmentation.forward :390-416 with ``mixers[i]`` standing in for layer i's Mamba
    (oracle.scan_ref.MambaRef carrying the same weights); DropPath is the identity (eval / rate 0)."""
    hidden, residual, feats = x + pos, None, []
    for i, layer in enumerate(blocks.layers):
        residual = hidden if residual is None else hidden + residual
        hidden = mixers[i](layer.norm(residual))
        if i in blocks.fetch_idx:
            feats.append(blocks.norm_f(hidden + residual))
    return feats


def seg_head(model, pts_bcn, cls_label, sorted_center, feature_list, io_dtype=None):
    """get_model.forward :761-787: everything after the block stack.  ``model`` supplies the parameters
    (norm, label_conv, propagation_0, convs1-3, bns1-2); pts_bcn (B,3,N); sorted_center (B,L,3);
    feature_list: the taps (B,L,C) each.  ``io_dtype``: restate the head as it runs under torch.autocast
    (``_autocast_ops``; LayerNorm and log_softmax are on autocast's fp32 list)."""
    conv_op, bn_op = _autocast_ops(io_dtype)
    B, _, N = pts_bcn.shape
    fl = [model.norm(x.float()).transpose(-1, -2).contiguous() for x in feature_list]
    x = torch.cat(fl, dim=1)
    x_max = torch.max(x, 2)[0]
    x_avg = torch.mean(x, 2)
    x_max_feature = x_max.view(B, -1).unsqueeze(-1).repeat(1, 1, N)
    x_avg_feature = x_avg.view(B, -1).unsqueeze(-1).repeat(1, 1, N)
    cls_label_one_hot = cls_label.view(B, 16, 1)
    lc, lb, lact = model.label_conv
    cls_label_feature = lact(bn_op(lb, conv_op(lc, cls_label_one_hot))).repeat(1, 1, N)
    x_global_feature = torch.cat((x_max_feature, x_avg_feature, cls_label_feature), 1)
    f_level_0 = feature_propagation(model.propagation_0, pts_bcn, sorted_center.transpose(-1, -2), pts_bcn, x,
                                    io_dtype=io_dtype)
    x = torch.cat((f_level_0, x_global_feature), 1)
    x = F.relu(bn_op(model.bns1, conv_op(model.convs1, x)))
    x = model.dp1(x)
    x = F.relu(bn_op(model.bns2, conv_op(model.convs2, x)))
    x = conv_op(model.convs3, x)
    x = F.log_softmax(x.float(), dim=1)
    return x.permute(0, 2, 1)
