"""CPU restatement of the reference's MAE pre-training data flow.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows models/point_mamba.py
  MaskMamba_2.forward, ``orders`` branch          :2440-2541  (permutation matrices, boolean-mask selection,
                                                               reverse concatenation, blocks, norm)
  Point_MAE_Mamba.forward, spectral method branch :3135-3215  (mask-token restore loop, decoder, masked-token
                                                               selection, increase_dim, Chamfer loss)
  pytorch3d.loss.chamfer_distance (absent wheel; published semantics: squared L2 nearest neighbour, mean over
  points, both directions summed, batch_reduction=None)        called at :3203
with the reference's own tensor operations (matmul with one-hot permutations, x[bool_mask], index assignment in
Python loops).  The mixers are passed in as callables.  No reference fixtures exist: parity unpinned.
"""
from __future__ import annotations

import torch


def permutation_matrices(orders, G):
    """(B,k,G) index orders -> (B,k,G,G) one-hot P with (P @ x)[i] = x[order[i]] (the form ``orders`` has at :2441)."""
    return torch.nn.functional.one_hot(orders, G).to(torch.float32)


def encoder_flow(tokens, pos, neighborhood, center, bool_masked_pos, P, reverse, blocks, norm):
    """:2440-2541.  tokens (B,G,C), pos (B,G,C), neighborhood (B,G,M,3), center (B,G,3), bool_masked_pos (B,G),
    P (B,k,G,G).  Returns the reference's 6 tensors (policy / mask_ratio dropped)."""
    batch_size, seq_len, C = tokens.size()
    sorted_bool_masked_pos = torch.matmul(P, bool_masked_pos.unsqueeze(1).unsqueeze(-1).float()).squeeze(-1).bool()
    sorted_full = torch.matmul(P, tokens.unsqueeze(1))
    sorted_tokens_t = sorted_full[~sorted_bool_masked_pos].view(batch_size, -1, C)
    sorted_pos_full = torch.matmul(P, pos.unsqueeze(1))
    sorted_pos_full_t = sorted_pos_full.flatten(1, 2)
    sorted_pos_t = sorted_pos_full[~sorted_bool_masked_pos].view(batch_size, -1, C)
    sorted_pos_mask_t = sorted_pos_full[sorted_bool_masked_pos].view(batch_size, -1, C)
    sorted_neighborhood_full = torch.einsum('bhij,bjkl->bhikl', P, neighborhood)
    sorted_neighborhood_t = sorted_neighborhood_full.flatten(1, 2)
    sorted_bool_masked_pos_list = list(torch.unbind(sorted_bool_masked_pos, dim=1))
    if reverse:
        x_vis = torch.cat((sorted_tokens_t, sorted_tokens_t.flip(1)), 1)
        p_vis = torch.cat((sorted_pos_t, sorted_pos_t.flip(1)), 1)
        sorted_pos_mask = torch.cat((sorted_pos_mask_t, sorted_pos_mask_t.flip(1)), 1)
        sorted_pos_full_out = torch.cat((sorted_pos_full_t, sorted_pos_full_t.flip(1)), 1)
        sorted_neighborhood = torch.cat((sorted_neighborhood_t, sorted_neighborhood_t.flip(1)), 1)
        sorted_bool_masked_pos_tensor = torch.cat(sorted_bool_masked_pos_list, -1).flip(-1)
    else:
        raise NotImplementedError("the reference only assigns its outputs when reverse == True (:2513-2532)")
    x_vis = norm(blocks(x_vis, p_vis))
    return (x_vis, sorted_bool_masked_pos_list, sorted_pos_mask, sorted_pos_full_out, sorted_bool_masked_pos_tensor,
            sorted_neighborhood)


def restore_tokens(x_vis, mask_token, sorted_bool_masked_pos_list, sorted_bool_masked_pos_tensor, n_masked, n_visible,
                   G):
    """:3150-3190: learnable mask tokens at the masked positions, visible tokens back at theirs."""
    B, _, D = x_vis.shape
    N = 2 * len(sorted_bool_masked_pos_list) * n_masked
    mt = mask_token.expand(B, N, -1)
    x_full_list = []
    for cnt, i in enumerate(sorted_bool_masked_pos_list):
        x_full = torch.zeros((B, G, D))
        mask_token_part = mt[:, (cnt * n_masked):(cnt + 1) * n_masked, :]
        x_vis_part = x_vis[:, (cnt * n_visible):(cnt + 1) * n_visible, :]
        mask_indices = torch.where(i == 1)
        vis_indices = torch.where(i == 0)
        x_full[mask_indices] = mask_token_part.reshape(-1, D)[0:len(mask_indices[0])]
        x_full[vis_indices] = x_vis_part.reshape(-1, D)[0:len(vis_indices[0])]
        x_full_list.append(x_full)
    cnt = len(sorted_bool_masked_pos_list)
    x_full_tensor_1 = torch.cat(x_full_list, 1)
    x_full_tensor_2 = torch.zeros_like(x_full_tensor_1)
    mask_token_part = mt[:, (cnt * n_masked):, :]
    x_vis_part = x_vis[:, (cnt * n_visible):, :]
    mask_indices = torch.where(sorted_bool_masked_pos_tensor == 1)
    vis_indices = torch.where(sorted_bool_masked_pos_tensor == 0)
    x_full_tensor_2[mask_indices] = mask_token_part.reshape(-1, D)[0:len(mask_indices[0])]
    x_full_tensor_2[vis_indices] = x_vis_part.reshape(-1, D)[0:len(vis_indices[0])]
    return torch.cat((x_full_tensor_1, x_full_tensor_2), 1)


def chamfer_distance(x, y):
    """pytorch3d.loss.chamfer_distance(x, y, batch_reduction=None)[0] for equal-length clouds:
    x (P,n,3), y (P,m,3) -> (P,)."""
    d = ((x.unsqueeze(2) - y.unsqueeze(1)) ** 2).sum(-1)          # (P,n,m) squared L2
    return d.min(dim=2)[0].mean(dim=1) + d.min(dim=1)[0].mean(dim=1)


def decoder_flow(x_vis, mask_token, masks, mask_tensor, sorted_pos_full, sorted_neighborhood, mask_ratio, G, decoder,
                 increase_dim):
    """:3135-3215 -> (loss, rebuild_points, gt_points, x_full)."""
    B, _, C = x_vis.shape
    n_masked = int(mask_ratio * G)
    n_visible = G - n_masked
    x_full = restore_tokens(x_vis, mask_token, masks, mask_tensor, n_masked, n_visible, G)
    x_rec = decoder(x_full, sorted_pos_full)
    final_mask = torch.cat((torch.cat(masks, 1), mask_tensor), 1)
    x_rec = x_rec[final_mask].reshape(B, -1, C)
    B, M, C = x_rec.shape
    rebuild = increase_dim(x_rec.transpose(1, 2)).transpose(1, 2).reshape(B * M, -1, 3)
    gt = sorted_neighborhood[final_mask].reshape(B * M, -1, 3)
    loss = chamfer_distance(rebuild.float(), gt.float()).mean()
    return loss, rebuild, gt, x_full
