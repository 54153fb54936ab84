"""Part-segmentation caller of the hot path: the reference's ``get_model``
(part_segmentation/models/pt_mamba.py:420-790; BASELINE config 5), with its parameter names and shapes.

What runs where
  * grouping (FPS kernel + k-NN), patch encoder (GEMMs + bn_relu / group_max kernels), spectral ordering
    (SAST :725-759 or HLT :665-723 through the spectral kernels), 12 Mamba blocks with taps after layers
    ``fetch_idx`` (``MixerModelForSegmentation`` :320-416) -- the same HIP path as the classifier;
  * feature propagation (pointnet2_utils.py:262-305): 3-nearest-centre interpolation on csrc/interp.hip, its
    1x1-conv MLP as GEMMs + bn_relu kernels;
  * the per-point head (:770-787) as GEMMs on token-major (B*N, C) tensors; every part of its input that is
    constant over the points of a sample (max / mean of the token features, the class-label embedding: 2368 of the
    3392 channels) enters as a per-sample additive term of the first BatchNorm instead of being repeated N times:
    cat([f, g.repeat(N)]) @ W^T == f @ W[:, :1024]^T + (g @ W[:, 1024:]^T)[sample].

Layout: the reference keeps (B, C, N) tensors for Conv1d; here everything stays token-major (B, N, C) and the
transposes disappear.  Conv1d(k=1) weights are used as (out, in) matrices, so a reference state_dict loads
unchanged.
"""
from __future__ import annotations

from functools import partial
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import spectral
from .add_norm import add_layer_norm_fn
from .block import DropPath, _init_weights, create_block
from .encoder_ops import bn_relu_fn, token_linear
from .interp import three_interpolate, three_nn
from .point_mamba import Encoder, Group


def default_seg_config(**over):
    """part_segmentation/cfgs/config.yaml."""
    cfg = dict(trans_dim=384, depth=12, drop_path_rate=0.1, rms_norm=False, drop_path=0.2, drop_out=0.,
               fetch_idx=(3, 7, 11), method="HLT", reverse=True, k_top_eigenvectors=4, smallest=True, knn_graph=10,
               symmetric=True, self_loop=True, alpha=10., binary=False, num_group=128, group_size=32)
    cfg.update(over)
    return SimpleNamespace(**cfg)


class MixerModelForSegmentation(nn.Module):
    """reference pt_mamba.py:320-416: the block stack, returning norm_f(hidden + residual) after each layer in
    ``fetch_idx`` (no DropPath on the tap)."""

    def __init__(self, d_model, n_layer, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False, initializer_cfg=None,
                 fused_add_norm=False, residual_in_fp32=False, drop_path=0.1, fetch_idx=(3, 7, 11), device=None,
                 dtype=None):
        super().__init__()
        if rms_norm:
            raise NotImplementedError("rms_norm=True needs mamba-ssm's Triton RMSNorm; the reference cfg sets False")
        kw = {"device": device, "dtype": dtype}
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.fetch_idx = tuple(fetch_idx)
        self.layers = nn.ModuleList([
            create_block(d_model, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         drop_path=drop_path, **kw) for i in range(n_layer)])
        self.norm_f = nn.LayerNorm(d_model, eps=norm_epsilon, **kw)
        self.apply(partial(_init_weights, n_layer=n_layer, **(initializer_cfg if initializer_cfg is not None else {})))
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, input_ids, pos, inference_params=None):
        hidden_states = input_ids + pos
        residual = None
        feats = []
        for i, layer in enumerate(self.layers):
            hidden_states, residual = layer(hidden_states, residual, inference_params=inference_params)
            if i in self.fetch_idx:
                if hidden_states.is_cuda:
                    feats.append(add_layer_norm_fn(hidden_states, residual, self.norm_f.weight, self.norm_f.bias,
                                                   self.norm_f.eps, out_dtype=self.norm_f.weight.dtype)[0])
                else:
                    feats.append(self.norm_f((hidden_states + residual).to(self.norm_f.weight.dtype)))
        return feats


class PointNetFeaturePropagation(nn.Module):
    """reference pointnet2_utils.py:262-311 (same parameter names); token-major interface:
    forward(xyz1 (B,N,3), xyz2 (B,S,3), points1 (B,N,D1) or None, points2 (B,S,D2)) -> (B*N, mlp[-1])."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out))
            last = out

    def forward(self, xyz1, xyz2, points1, points2):
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interp = points2.expand(-1, N, -1)
        else:
            idx, w = three_nn(xyz1, xyz2)
            interp = three_interpolate(points2, idx, w)
        interp = interp.reshape(B * N, -1)
        w0 = self.mlp_convs[0].weight.squeeze(-1)
        if points1 is not None:
            # cat([points1, interp]) @ W^T as two products accumulated in place (no (B*N, D1+D2) tensor)
            d1 = points1.shape[-1]
            h = F.linear(points1.reshape(B * N, d1).to(interp.dtype), w0[:, :d1], self.mlp_convs[0].bias)
            h = torch.addmm(h, interp, w0[:, d1:].t().to(interp.dtype))
        else:
            h = token_linear(interp, w0, self.mlp_convs[0].bias)
        h = bn_relu_fn(h, self.mlp_bns[0])
        for conv, bn in zip(list(self.mlp_convs)[1:], list(self.mlp_bns)[1:]):
            h = bn_relu_fn(token_linear(h, conv.weight.squeeze(-1), conv.bias), bn)
        return h


class PartSegMamba(nn.Module):
    """The reference's ``get_model`` (pt_mamba.py:420): forward(pts (B,3,N), cls_label (B,16)) -> (B,N,cls_dim)
    log-probabilities."""

    def __init__(self, cls_dim, config=None):
        super().__init__()
        config = default_seg_config() if config is None else config
        self.config = config
        self.trans_dim = config.trans_dim
        self.depth = config.depth
        self.cls_dim = cls_dim
        self.group_size = getattr(config, "group_size", 32)
        self.num_group = getattr(config, "num_group", 128)
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.encoder_dims = 384 if self.trans_dim == 384 else self.trans_dim
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        self.blocks = MixerModelForSegmentation(d_model=self.trans_dim, n_layer=self.depth, rms_norm=config.rms_norm,
                                                drop_path=config.drop_path, fetch_idx=config.fetch_idx)
        self.drop_out = nn.Dropout(getattr(config, "drop_out", 0.))
        self.drop_path_rate = config.drop_path_rate
        self.drop_path_block = DropPath(self.drop_path_rate) if self.drop_path_rate > 0. else nn.Identity()
        self.norm = nn.LayerNorm(self.trans_dim)
        self.label_conv = nn.Sequential(nn.Conv1d(16, 64, kernel_size=1, bias=False), nn.BatchNorm1d(64),
                                        nn.LeakyReLU(0.2))
        nf = len(tuple(config.fetch_idx)) * self.trans_dim                     # 1152
        self.propagation_0 = PointNetFeaturePropagation(in_channel=nf + 3, mlp=[self.trans_dim * 4, 1024])
        self.convs1 = nn.Conv1d(1024 + 2 * nf + 64, 512, 1)                    # 3392 at the reference sizes
        self.dp1 = nn.Dropout(0.5)
        self.convs2 = nn.Conv1d(512, 256, 1)
        self.convs3 = nn.Conv1d(256, self.cls_dim, 1)
        self.bns1 = nn.BatchNorm1d(512)
        self.bns2 = nn.BatchNorm1d(256)
        self.relu = nn.ReLU()
        self.method = config.method
        self.reverse = config.reverse
        self.k_top_eigenvectors = config.k_top_eigenvectors
        self.smallest = config.smallest
        self.knn_graph = config.knn_graph
        self.symmetric = config.symmetric
        self.self_loop = config.self_loop
        self.alpha = config.alpha
        self.binary = config.binary
        if self.method not in ("HLT", "SAST", "Point_MAMBA"):
            raise NotImplementedError(f"ordering method {self.method!r}")
        self.hlt_rand = True          # the reference's torch.rand tie-break (:655); tests switch it off

    # ---- token ordering ------------------------------------------------------------------------------------------
    def _spectral(self, center):
        """What the ordering needs from the eigen-solver: eigenvectors (HLT) or the k argsorts (SAST); None otherwise.
        Depends on the centres only, so forward() runs it on a side stream underneath the patch encoder."""
        if self.method == "Point_MAMBA":
            return None
        adj = spectral.create_graph_from_centers(center, self.knn_graph, self.alpha, self.symmetric, self.self_loop,
                                                 self.binary)
        if self.method == "HLT":                                                # :667-668
            return spectral._eig(adj, self.k_top_eigenvectors, self.smallest, False, want_all=False)[1]
        return spectral._eig(adj, self.k_top_eigenvectors, self.smallest, False, want_all=False, want_order=True)[4]

    def order_tokens(self, tokens, pos, center, spec=None):
        """-> (tokens, pos, center) in sequence order: (B, L, C), (B, L, C), (B, L, 3)."""
        if self.method == "Point_MAMBA":                                        # :640-663
            idx = torch.cat([center[:, :, a].argsort(dim=-1) for a in range(3)], dim=1)
            ex = idx.unsqueeze(-1)
            return (torch.gather(tokens, 1, ex.expand(-1, -1, tokens.shape[-1])),
                    torch.gather(pos, 1, ex.expand(-1, -1, pos.shape[-1])),
                    torch.gather(center, 1, ex.expand(-1, -1, 3)))
        if spec is None:
            spec = self._spectral(center)
        if self.method == "HLT":                                                # :665-723
            rand = torch.rand(center.shape[0], center.shape[1], device=center.device) if self.hlt_rand else None
            t, p, c, _ = spectral.hlt_assemble(tokens, pos, center, spec, self.k_top_eigenvectors, rand=rand)
            return t, p, c
        # SAST :725-759 (the graph is create_graph_from_centers here, not the feature-space variant)
        idx = spectral.sast_index_map(spec, self.reverse)
        ex = idx.unsqueeze(-1)
        return (torch.gather(tokens, 1, ex.expand(-1, -1, tokens.shape[-1])),
                torch.gather(pos, 1, ex.expand(-1, -1, pos.shape[-1])),
                torch.gather(center, 1, ex.expand(-1, -1, 3)))

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, pts, cls_label):
        B, _, N = pts.shape
        pts = pts.transpose(-1, -2).contiguous()                                # (B, N, 3)
        neighborhood, center, _ = self.group_divider(pts)
        spec = spectral.run_on_side_stream(lambda: self._spectral(center), center)
        tokens = self.encoder(neighborhood)
        pos = self.pos_embed(center)
        x, spos, scenter = self.order_tokens(tokens, pos, center, spec())
        feats = self.blocks(x, spos)
        feats = torch.cat([self.norm(f) for f in feats], dim=-1)                # (B, L, 1152)
        x_max = feats.max(dim=1)[0]
        x_avg = feats.mean(dim=1)
        label = self.label_conv(cls_label.view(B, 16, 1).to(feats.dtype)).squeeze(-1)        # (B, 64)
        glob = torch.cat([x_max, x_avg, label.to(x_max.dtype)], dim=1)          # (B, 2368)
        f0 = self.propagation_0(pts, scenter, pts, feats)                        # (B*N, 1024)
        w1 = self.convs1.weight.squeeze(-1)
        nf0 = f0.shape[1]
        gterm = F.linear(glob, w1[:, nf0:], self.convs1.bias)                   # (B, 512)
        h = token_linear(f0, w1[:, :nf0])
        if N % 256 == 0 or 256 % N == 0:
            h = bn_relu_fn(h, self.bns1, gterm=gterm, group=N)
        else:
            h = bn_relu_fn(h + gterm.repeat_interleave(N, dim=0).to(h.dtype), self.bns1)
        h = self.dp1(h)
        h = bn_relu_fn(token_linear(h, self.convs2.weight.squeeze(-1), self.convs2.bias), self.bns2)
        h = token_linear(h, self.convs3.weight.squeeze(-1), self.convs3.bias)
        return F.log_softmax(h.float(), dim=-1).view(B, N, self.cls_dim)


get_model = PartSegMamba          # the reference's class name


class get_loss(nn.Module):
    """reference pt_mamba.py:790-796."""

    def forward(self, pred, target):
        return F.nll_loss(pred, target)
