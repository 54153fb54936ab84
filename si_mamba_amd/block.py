"""Host-side mirror of the reference's operator API around the mixer.

  Block          reference models/block.py:17-76   (Add -> LayerNorm -> mixer, DropPath)
  create_block   reference models/point_mamba.py:147-175
  _init_weights  reference models/point_mamba.py:115-144
  MixerModel     reference models/point_mamba.py:178-272

Same constructor arguments, forward signatures, return values and parameter names, so a
state_dict of the reference's ``blocks.*`` loads unchanged.  The Triton fused add+norm path
(``fused_add_norm=True``) is never enabled by any reference config and is refused here
(no Triton in this framework).
"""
from __future__ import annotations

import math
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .add_norm import add_layer_norm_fn
from .mamba_simple import Mamba
from .out_norm import out_proj_add_ln_fn, out_proj_add_ln_ok


class DropPath(nn.Module):
    """Per-sample stochastic depth (the timm layer the reference imports at models/block.py:13)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def rowscale(self, x):
        """Per-sample factor (B,) this layer would multiply x by, or None when it is the identity."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        mask = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask

    def forward(self, x):
        mask = self.rowscale(x)
        return x if mask is None else x * mask.to(x.dtype).view((-1,) + (1,) * (x.dim() - 1))


class Block(nn.Module):
    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False,
                 residual_in_fp32=False, drop_path=0.):
        super().__init__()
        if fused_add_norm:
            raise NotImplementedError("fused_add_norm (Triton) is not part of this framework; "
                                      "no reference config enables it")
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = False
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, inference_params=None, A=None):
        """hidden_states = Mixer(LN(residual)); returns (hidden_states, residual).  ``A`` (optional, specific to this
        implementation): the mixer's -exp(A_log) when the caller has already formed it (MixerModel does, for all
        layers in one launch); handed on as an argument, never parked on the module.

        Add (+ DropPath) + LayerNorm run as one HIP pass each way (add_norm.py) for LayerNorm blocks on the
        GPU; the composed torch form below is the reference's own and serves any other norm / device."""
        if hidden_states.is_cuda and type(self.norm) is nn.LayerNorm and hidden_states.dim() == 3:
            scale = self.drop_path.rowscale(hidden_states) if isinstance(self.drop_path, DropPath) else None
            hidden_states, residual = add_layer_norm_fn(hidden_states, residual, self.norm.weight, self.norm.bias,
                                                        self.norm.eps, rowscale=scale)
        else:
            residual = (self.drop_path(hidden_states) + residual) if residual is not None else hidden_states
            hidden_states = self.norm(residual.to(dtype=self.norm.weight.dtype))
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
        if A is not None:
            hidden_states = self.mixer(hidden_states, inference_params=inference_params, A=A)
        else:
            hidden_states = self.mixer(hidden_states, inference_params=inference_params)
        return hidden_states, residual

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True,
                  n_residuals_per_layer=1):
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        # GPT-2 style: residual-branch output projections scaled by 1/sqrt(#residual layers)
        for name, p in module.named_parameters():
            if name in ("out_proj.weight", "fc2.weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, drop_path=0., device=None, dtype=None):
    if rms_norm:
        raise NotImplementedError("rms_norm=True needs mamba-ssm's Triton RMSNorm; every reference cfg sets False")
    ssm_cfg = {} if ssm_cfg is None else ssm_cfg
    factory_kwargs = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, layer_idx=layer_idx, **ssm_cfg, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm, eps=norm_epsilon, **factory_kwargs)
    block = Block(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32, drop_path=drop_path)
    block.layer_idx = layer_idx
    return block


class MixerModel(nn.Module):
    def __init__(self, d_model: int, n_layer: int, ssm_cfg=None, norm_epsilon: float = 1e-5,
                 rms_norm: bool = False, initializer_cfg=None, fused_add_norm=False,
                 residual_in_fp32=False, drop_out_in_block: float = 0., drop_path: float = 0.1,
                 device=None, dtype=None) -> None:
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.layers = nn.ModuleList([
            create_block(d_model, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         drop_path=drop_path, **factory_kwargs)
            for i in range(n_layer)])
        self.norm_f = nn.LayerNorm(d_model, eps=norm_epsilon, **factory_kwargs)
        self.apply(partial(_init_weights, n_layer=n_layer,
                           **(initializer_cfg if initializer_cfg is not None else {})))
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.drop_out_in_block = nn.Dropout(drop_out_in_block) if drop_out_in_block > 0. else nn.Identity()

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return {i: layer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)
                for i, layer in enumerate(self.layers)}

    def _first_block_on_distinct_tokens(self, tokens, pos, token_index, A=None, emit_y=False):
        """Block 0 when the sequence is ``gather(tokens + pos, token_index)``: Add + LayerNorm + in_proj on the G
        distinct tokens, expanded to the L positions by a copy kernel (seq_expand.py).  -> (hidden, residual) of
        block 0 as the reference's Block.forward returns them (``emit_y``: the mixer's output before out_proj in
        place of hidden, see _chain), or None when the route does not apply."""
        from . import seq_expand
        layer = self.layers[0]
        mixer = layer.mixer
        if not (tokens.is_cuda and type(layer.norm) is nn.LayerNorm and isinstance(mixer, Mamba)
                and mixer.use_fast_path and seq_expand.expansion_ok(tokens, token_index)):
            return None
        inv32 = seq_expand.inverse_positions(token_index, tokens.shape[1])
        if inv32 is None:
            return None
        idx32 = token_index.to(torch.int32).contiguous()
        # residual_0 = tokens + pos and its LayerNorm in one pass over the G tokens.  (Low-precision operands: the
        # reference's `input_ids + pos` rounds the sum to that dtype first -- keep that rounding.)
        if tokens.dtype == torch.float32:
            normed, res0 = add_layer_norm_fn(tokens, pos, layer.norm.weight, layer.norm.bias, layer.norm.eps)
        else:
            normed, res0 = add_layer_norm_fn(tokens + pos, None, layer.norm.weight, layer.norm.bias, layer.norm.eps)
        xz = seq_expand.seq_gather_last(mixer.in_proj_xz(normed), idx32, inv32)          # (B, 2D, L)
        hidden = mixer.forward_xz(xz, A=A, emit_y=emit_y)
        residual = torch.gather(res0, 1, token_index.unsqueeze(-1).expand(-1, -1, res0.shape[-1]))
        return hidden, residual

    def _precompute_A(self):
        """A = -exp(A_log) of every layer in three launches (stack, exp, neg) instead of two per layer, and one
        exp / neg backward for all of them.  -> per-layer list (slices of one tensor) handed to the blocks as an
        ARGUMENT, or a list of None when the layers are not uniform Mamba mixers on the GPU (each then forms its own)."""
        mixers = [layer.mixer for layer in self.layers]
        if (len(mixers) > 1 and all(type(m) is Mamba for m in mixers) and mixers[0].A_log.is_cuda
                and all(m.A_log.shape == mixers[0].A_log.shape for m in mixers)):
            return list((-torch.exp(torch.stack([m.A_log for m in mixers]).float())).unbind(0))
        return [None] * len(mixers)

    def _chain_ok(self, x):
        """Can the stack run as  add+LN -> [in_proj -> mixer body -> (out_proj + add + LN)] * n  with the bracketed
        out_proj / add / LayerNorm as ONE kernel (out_norm.py)?  bf16 compute (autocast or bf16 modules), plain
        LayerNorm blocks around fast-path Mamba mixers without an out_proj bias, nothing between the blocks."""
        if not (_lib.fuse_out_norm_enabled() and x.is_cuda and x.dim() == 3
                and isinstance(self.drop_out_in_block, nn.Identity) and type(self.norm_f) is nn.LayerNorm):
            return False
        # the dtype the projections compute in: autocast's, else the parameters'
        io = (torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda")
              else self.layers[0].mixer.in_proj.weight.dtype)
        if io != torch.bfloat16:
            return False
        d_model = x.shape[-1]
        for layer in self.layers:
            m = layer.mixer
            if not (type(layer) is Block and type(layer.norm) is nn.LayerNorm and type(m) is Mamba and m.use_fast_path
                    and m.out_proj.bias is None and m.d_model == d_model and d_model % 128 == 0 and d_model <= 384
                    and m.d_inner % 64 == 0):
                return False
        return True

    def _chain(self, first, y, normed, residual, A_all):
        """Blocks ``first``.. of the stack and its final norm on the fused route.  Enters either with ``normed`` (the
        LayerNorm output block ``first`` feeds its mixer) or with ``y`` (block ``first - 1``'s mixer output before
        out_proj); ``residual`` is the fp32 residual stream that goes with it.  What the reference computes across each
        block boundary -- out_proj (models/block.py:72), residual = drop_path(hidden) + residual and norm(residual) at
        the top of the next block (:56-58), or norm_f after the last (models/point_mamba.py:257-258) -- is one kernel;
        shapes it does not take run the same three ops through the library GEMM and the add + LayerNorm kernel."""
        n = len(self.layers)
        i = first
        while True:
            if y is not None:                               # close block i - 1, open block i (or finish the stack)
                prev = self.layers[i - 1].mixer
                if i < n:
                    norm, dp = self.layers[i].norm, self.layers[i].drop_path
                    scale = dp.rowscale(y) if isinstance(dp, DropPath) else None
                    out_dtype = y.dtype                     # what add_layer_norm_fn hands a block's in_proj
                else:
                    norm, scale, out_dtype = self.norm_f, None, self.norm_f.weight.dtype
                if out_proj_add_ln_ok(y, prev.out_proj.weight, prev.d_model):
                    _lib.count("out_proj_add_ln")
                    normed, residual = out_proj_add_ln_fn(y, prev.out_proj.weight, residual, norm.weight, norm.bias,
                                                          norm.eps, rowscale=scale, out_dtype=out_dtype)
                else:
                    hidden = torch.matmul(y.transpose(1, 2), prev.out_proj.weight.t().to(y.dtype))
                    normed, residual = add_layer_norm_fn(hidden, residual, norm.weight, norm.bias, norm.eps,
                                                         rowscale=scale, out_dtype=out_dtype)
                if i == n:
                    return normed
            mixer = self.layers[i].mixer
            y = mixer.forward_xz(mixer.in_proj_xz(normed), A=A_all[i], emit_y=True)
            i += 1

    def forward(self, input_ids, pos, inference_params=None, token_index=None, balanced_index=False):
        """Reference signature (models/point_mamba.py:247).  ``token_index`` (B, L) int64, optional and specific to
        this implementation: when given, ``input_ids`` / ``pos`` are the G DISTINCT tokens (B, G, C) and the sequence
        the reference would be handed is their gather by ``token_index``; the result is the same.
        ``balanced_index=True`` is the caller's statement that every token occurs exactly L / G times in every row of
        ``token_index`` (a concatenation of permutations, which is what the SAST / MAMBA assemblies are): only then
        does the first block's per-token head run on the G tokens (the adjoint of that expansion sums a fixed number
        of positions per token); any other index takes the reference's route on the gathered sequence."""
        first = 0
        residual = None
        A_all = self._precompute_A()
        chain = inference_params is None and len(self.layers) > 0 and self._chain_ok(input_ids)
        if token_index is not None:
            done = (self._first_block_on_distinct_tokens(input_ids, pos, token_index, A=A_all[0], emit_y=chain)
                    if (balanced_index and inference_params is None) else None)
            if done is not None:
                hidden_states, residual = done
                if chain:                                   # hidden_states is block 0's y (B, d_inner, L)
                    return self._chain(1, hidden_states, None, residual, A_all)
                hidden_states = self.drop_out_in_block(hidden_states)
                first = 1
            else:                                       # the reference's route on the expanded sequence
                ex = token_index.unsqueeze(-1)
                input_ids = torch.gather(input_ids, 1, ex.expand(-1, -1, input_ids.shape[-1]))
                pos = torch.gather(pos, 1, ex.expand(-1, -1, pos.shape[-1]))
        if first == 0:
            hidden_states = input_ids + pos
        if chain:
            l0 = self.layers[0]
            scale = l0.drop_path.rowscale(hidden_states) if isinstance(l0.drop_path, DropPath) else None
            normed, residual = add_layer_norm_fn(hidden_states, None, l0.norm.weight, l0.norm.bias, l0.norm.eps,
                                                 rowscale=scale)
            return self._chain(0, None, normed, residual, A_all)
        for i in range(first, len(self.layers)):
            hidden_states, residual = self.layers[i](hidden_states, residual, inference_params=inference_params,
                                                     A=A_all[i])
            hidden_states = self.drop_out_in_block(hidden_states)
        if hidden_states.is_cuda and type(self.norm_f) is nn.LayerNorm and hidden_states.dim() == 3:
            # the stack's output norm returns the parameter dtype (fp32) under autocast too, as F.layer_norm does
            # there; the block norms feed a GEMM and may hand over the autocast dtype directly
            return add_layer_norm_fn(hidden_states, residual, self.norm_f.weight, self.norm_f.bias,
                                     self.norm_f.eps, out_dtype=self.norm_f.weight.dtype)[0]
        residual = (hidden_states + residual) if residual is not None else hidden_states
        return self.norm_f(residual.to(dtype=self.norm_f.weight.dtype))
