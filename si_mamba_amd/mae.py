"""MAE pre-training caller of the hot path: the reference's ``Point_MAE_Mamba`` with
``method == "smallest_eigenvectors_seperate_learnable_tokens"`` (models/point_mamba.py:2870-3219, encoder
``MaskMamba_2`` :2135-2541, decoder ``MambaDecoder_SST`` :2836-2866; BASELINE config 4), parameter names kept.

Token orders.  ``MaskMamba_2.forward`` consumes ``orders``: k permutations of the G patches per sample (:2440-2500,
as permutation matrices ``P``; sorted = P @ x).  The committed runner fills them from a wavelet research module
(``use_wavelets=True``, tools/runner_pretrain.py:244) that SURVEY.md marks out of scope; the spectral orders of the
published method -- ascending argsort of the k smallest Laplacian eigenvectors, the commented block at
:2343-2393 and what the classifier uses -- are what this module feeds through the same data flow
(``spectral.spectral_order`` on the HIP kernels).  Any (B, k, G) int64 ``orders`` tensor can be passed instead.

Everything the reference does with boolean-mask indexing and Python loops over the k orderings (:2440-2530,
:3150-3190) is index arithmetic here: one stable sort of the sorted mask gives the visible / masked positions of
every ordering, then gathers and one scatter.  Same tensors, same order of tokens.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, spectral
from .block import MixerModel
from .point_mamba import Encoder, Group


def default_mae_config(**over):
    """cfgs/pretrain.yaml:37-68."""
    tc = dict(mask_ratio=0.6, mask_type="rand", trans_dim=384, encoder_dims=384, depth=12, drop_path_rate=0.1,
              num_heads=6, decoder_depth=4, decoder_num_heads=6,
              method="smallest_eigenvectors_seperate_learnable_tokens", reverse=True, knn_graph=20,
              k_top_eigenvectors=4, smallest=True, alpha=10, symmetric=True, self_loop=False, binary=True)
    top = dict(group_size=32, num_group=64, loss="cdl2", rms_norm=False, use_cls_token=False, drop_path=0.1,
               drop_out=0.1)
    for k, v in over.items():
        (tc if k in tc else top)[k] = v
    top["transformer_config"] = SimpleNamespace(**tc)
    return SimpleNamespace(**top)


# ---- Chamfer distance on the HIP kernels ----------------------------------------------------------------------------
class ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        _lib.require_gpu(pred, "chamfer_distance")
        lib = _lib.load()
        p = pred.float().contiguous()
        g = gt.detach().float().contiguous()
        pairs, n, _ = p.shape
        m = g.shape[1]
        dev = p.device
        dist = torch.empty(pairs, device=dev, dtype=torch.float32)
        i1 = torch.empty(pairs, n, device=dev, dtype=torch.uint8)
        i2 = torch.empty(pairs, m, device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            rc = lib.simamba_chamfer_fwd(p.data_ptr(), g.data_ptr(), dist.data_ptr(), i1.data_ptr(), i2.data_ptr(),
                                         pairs, n, m, _lib.stream_ptr(dev))
        _lib.check(rc, "simamba_chamfer_fwd")
        ctx.save_for_backward(p, g, i1, i2)
        ctx.in_dtype = pred.dtype
        return dist

    @staticmethod
    def backward(ctx, ddist):
        p, g, i1, i2 = ctx.saved_tensors
        lib = _lib.load()
        pairs, n, _ = p.shape
        dp = torch.empty_like(p)
        dd = ddist.float().contiguous()
        with torch.cuda.device(p.device):
            rc = lib.simamba_chamfer_bwd(p.data_ptr(), g.data_ptr(), dd.data_ptr(), i1.data_ptr(), i2.data_ptr(),
                                         dp.data_ptr(), pairs, n, g.shape[1], _lib.stream_ptr(p.device))
        _lib.check(rc, "simamba_chamfer_bwd")
        return dp.to(ctx.in_dtype), None


def chamfer_distance(pred, gt):
    """(pairs, n, 3), (pairs, m, 3) -> (pairs,): pytorch3d ``chamfer_distance(..., batch_reduction=None)[0]``."""
    return ChamferFn.apply(pred, gt)


# ---- index arithmetic shared by encoder and decoder ---------------------------------------------------------------
def masked_positions(orders, mask, nm=None):
    """orders (B,k,G) int64, mask (B,G) bool with the same number ``nm`` of True per row (pass it to avoid a host
    read-back) ->
    vis_pos (B,k,G-nm), msk_pos (B,k,nm): positions inside each ordered sequence, ascending."""
    smask = torch.gather(mask.unsqueeze(1).expand(-1, orders.shape[1], -1), 2, orders)        # (B,k,G)
    if nm is None:
        nm = int(mask[0].sum())                                                                  # host sync
    pos = torch.sort(smask.to(torch.uint8), dim=2, stable=True)[1]                               # False first
    G = orders.shape[2]
    return pos[:, :, :G - nm], pos[:, :, G - nm:], smask


def sequence_positions(pos_in_order, G, reverse):
    """(B,k,n) positions inside each ordering -> (B, k*n*(1+reverse)) positions inside the (1+reverse)*k*G token
    sequence [order 0 | ... | order k-1 | the same flipped], ascending -- the order in which the reference's
    boolean-mask indexing enumerates them."""
    B, k, n = pos_in_order.shape
    first = (pos_in_order + torch.arange(k, device=pos_in_order.device).view(1, k, 1) * G).reshape(B, k * n)
    if not reverse:
        return first
    return torch.cat([first, (2 * k * G - 1) - first.flip(1)], dim=1)


class MaskMamba_2(nn.Module):
    """MAE encoder, reference :2135-2541."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        tc = config.transformer_config
        self.mask_ratio = tc.mask_ratio
        self.group_size = config.group_size
        self.num_group = config.num_group
        self.trans_dim = tc.trans_dim
        self.depth = tc.depth
        self.k_top_eigenvectors = tc.k_top_eigenvectors
        self.encoder_dims = tc.encoder_dims
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.mask_type = tc.mask_type
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        self.blocks = MixerModel(d_model=self.trans_dim, n_layer=self.depth, rms_norm=config.rms_norm)
        self.norm = nn.LayerNorm(self.trans_dim)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):                                               # :2189-2200
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _mask_center_rand(self, center, noaug=False, generator=None):   # :2232-2255 (torch RNG instead of numpy)
        B, G, _ = center.shape
        if noaug or self.mask_ratio == 0:
            return torch.zeros(B, G, dtype=torch.bool, device=center.device)
        self.num_mask = int(self.mask_ratio * G)
        r = torch.rand(B, G, generator=generator, device="cpu" if generator is not None else center.device)
        rank = r.to(center.device).argsort(dim=1).argsort(dim=1)
        return rank < self.num_mask

    def _mask_center_block(self, center, noaug=False, generator=None):  # :2202-2230
        B, G, _ = center.shape
        if noaug or self.mask_ratio == 0:
            return torch.zeros(B, G, dtype=torch.bool, device=center.device)
        seed = torch.randint(0, G, (B,), generator=generator).to(center.device)
        d = (center - center[torch.arange(B, device=center.device), seed].unsqueeze(1)).norm(dim=-1)
        rank = d.argsort(dim=1).argsort(dim=1)
        return rank < int(self.mask_ratio * G)

    def forward(self, neighborhood, center, orders, reverse=True, noaug=False, mask=None):
        """-> dict with x_vis (B, n_vis_tokens, C) after blocks + norm and the index tensors the decoder needs."""
        B, G, _ = center.shape
        user_mask = mask is not None
        if mask is None:
            mask = (self._mask_center_rand if self.mask_type == "rand" else self._mask_center_block)(center, noaug)
        tokens = self.encoder(neighborhood)
        pos = self.pos_embed(center)
        if callable(orders):                       # still in flight on the side stream (Point_MAE_Mamba.forward)
            orders = orders()
        C = tokens.shape[-1]
        k = orders.shape[1]
        # number of masked patches per cloud: known on the host for the generated masks (no device read-back, so the
        # step stays capturable in a hipGraph); counted only for caller-supplied masks
        nm = None if user_mask else (0 if (noaug or self.mask_ratio == 0) else int(self.mask_ratio * G))
        vis_pos, msk_pos, smask = masked_positions(orders, mask, nm)
        vis_src = torch.gather(orders, 2, vis_pos).reshape(B, -1)                  # patch ids, visible, by ordering
        msk_src = torch.gather(orders, 2, msk_pos).reshape(B, -1)
        full_src = orders.reshape(B, k * G)
        if reverse:
            vis_src = torch.cat([vis_src, vis_src.flip(1)], 1)
            msk_src = torch.cat([msk_src, msk_src.flip(1)], 1)
            full_src = torch.cat([full_src, full_src.flip(1)], 1)

        def take(x, idx):
            return torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, x.shape[-1]))

        x_vis = self.blocks(take(tokens, vis_src), take(pos, vis_src))
        x_vis = self.norm(x_vis)
        return dict(x_vis=x_vis, orders=orders, pos_mask=take(pos, msk_src), pos_full=take(pos, full_src), mask=mask,
                    sorted_mask=smask, vis_seq=sequence_positions(vis_pos, G, reverse),
                    msk_seq=sequence_positions(msk_pos, G, reverse), msk_src=msk_src, C=C)


class MambaDecoder_SST(nn.Module):
    """reference :2836-2866."""

    def __init__(self, embed_dim=384, depth=4, norm_layer=nn.LayerNorm, config=None):
        super().__init__()
        self.blocks = MixerModel(d_model=embed_dim, n_layer=depth, rms_norm=config.rms_norm,
                                 drop_path=config.drop_path)
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x, pos, return_token_num=None):
        return self.head(self.norm(self.blocks(x, pos)))


class Point_MAE_Mamba(nn.Module):
    """reference :2870-3219, eigenvector-order route; forward(pts (B,N,3)) -> scalar Chamfer-L2 loss."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        tc = config.transformer_config
        if tc.method != "smallest_eigenvectors_seperate_learnable_tokens":
            raise NotImplementedError("only the spectral route of Point_MAE_Mamba is built")
        self.trans_dim = tc.trans_dim
        self.MAE_encoder = MaskMamba_2(config)
        self.group_size = config.group_size
        self.num_group = config.num_group
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.decoder_pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        self.decoder_depth = tc.decoder_depth
        self.MAE_decoder = MambaDecoder_SST(embed_dim=self.trans_dim, depth=self.decoder_depth, config=config)
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.increase_dim = nn.Sequential(nn.Conv1d(self.trans_dim, 3 * self.group_size, 1))
        nn.init.trunc_normal_(self.mask_token, std=.02)
        if config.loss not in ("cdl1", "cdl2"):
            raise NotImplementedError(config.loss)
        self.loss = config.loss
        self.method = tc.method
        self.reverse = tc.reverse
        self.k_top_eigenvectors = tc.k_top_eigenvectors
        self.smallest = tc.smallest
        self.knn_graph = tc.knn_graph
        self.alpha = tc.alpha
        self.symmetric = tc.symmetric
        self.self_loop = tc.self_loop
        self.binary = tc.binary

    def spectral_orders(self, center):
        """:3097 graph (create_graph_from_centers) -> k smallest eigenvectors -> ascending argsort, (B,k,G)."""
        adj = spectral.create_graph_from_centers(center, self.knn_graph, self.alpha, self.symmetric, self.self_loop,
                                                 self.binary)
        return spectral._eig(adj, self.k_top_eigenvectors, self.smallest, False, want_all=False, want_order=True)[4]

    def forward(self, pts, noaug=False, vis=False, tau=None, use_wavelets=False, use_diff_sort=False,
                ret_policy=False, ret_only_policy=False, save_pts_dir=None, epoch=None, orders=None, mask=None,
                return_parts=False, **kwargs):
        """Reference signature (models/point_mamba.py:3053-3054; the pre-training runner calls
        ``base_model(points, tau=tau, ret_policy=False, use_wavelets=True)``, tools/runner_pretrain.py:244).
        Built: the spectral-order route of the published method.  The wavelet-traversal, learned-permutation,
        differentiable-sort, policy and visualisation branches are outside the hot-path scope (SURVEY.md section
        2) and are refused by name instead of being silently replaced by the spectral orders.  ``orders`` / ``mask``
        / ``return_parts`` are this package's test hooks; ``save_pts_dir`` / ``epoch`` only feed the reference's
        point dumps and are ignored."""
        for name, on in (("tau", tau is not None), ("use_wavelets", use_wavelets), ("use_diff_sort", use_diff_sort),
                         ("ret_policy", ret_policy), ("ret_only_policy", ret_only_policy), ("vis", vis)):
            if on:
                raise NotImplementedError(f"Point_MAE_Mamba.forward({name}=...): that branch of reference "
                                          "models/point_mamba.py:3053-3219 is outside the SI-Mamba hot-path scope; "
                                          "the spectral-order route runs with the argument left at its default")
        if kwargs:
            raise TypeError(f"Point_MAE_Mamba.forward: unexpected arguments {sorted(kwargs)}")
        neighborhood, center, _ = self.group_divider(pts)
        B, G, M, _ = neighborhood.shape
        if orders is None:
            # the eigen-ordering only needs the centres: it runs on a side stream underneath the patch encoder
            orders = spectral.run_on_side_stream(lambda: self.spectral_orders(center), center)
        enc = self.MAE_encoder(neighborhood, center, orders, self.reverse, noaug, mask=mask)
        orders = enc["orders"]
        k = orders.shape[1]
        x_vis, C = enc["x_vis"], enc["C"]
        if noaug:
            return x_vis
        T = k * G * (2 if self.reverse else 1)
        # mask tokens at the masked positions, visible tokens back at theirs (:3150-3190)
        x_full = self.mask_token.to(x_vis.dtype).expand(B, T, C).scatter(
            1, enc["vis_seq"].unsqueeze(-1).expand(-1, -1, C), x_vis)
        x_rec = self.MAE_decoder(x_full, enc["pos_full"], enc["pos_mask"].shape[1])
        x_rec = torch.gather(x_rec, 1, enc["msk_seq"].unsqueeze(-1).expand(-1, -1, C))          # (B, Mtok, C)
        Mtok = x_rec.shape[1]
        w = self.increase_dim[0]
        rebuild = F.linear(x_rec.reshape(B * Mtok, C), w.weight.squeeze(-1), w.bias).reshape(B * Mtok, -1, 3)
        gt = torch.gather(neighborhood, 1, enc["msk_src"].view(B, Mtok, 1, 1).expand(-1, -1, M, 3))
        gt = gt.reshape(B * Mtok, M, 3)
        loss = chamfer_distance(rebuild.float(), gt.float()).mean()
        if return_parts:
            return loss, dict(enc, rebuild=rebuild, gt=gt, x_full=x_full)
        return loss
