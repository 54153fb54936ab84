"""Shared pieces of the model-level parity tests: the oracle composition of the reference's block stack.

TEST INFRASTRUCTURE.  ``oracle_stack`` restates MixerModel.forward (reference models/point_mamba.py:247-258) and
Block.forward (models/block.py:47-73) around CPU oracle mixers (oracle.scan_ref.MambaRef) that carry the device
model's weights.  ``io_dtype=torch.bfloat16`` restates the same stack as it runs under ``torch.autocast`` on the
reference's CUDA path: LayerNorm computes and returns fp32 (autocast's fp32 list), the residual stream is fp32
(``residual = drop_path(hidden) + residual`` promotes), every mixer op reads and writes bf16 with fp32
accumulation (MambaRef.forward's ``io_dtype``), and ``input_ids + pos`` follows torch's type promotion (a bf16 add
when both operands arrive as bf16).
"""
import torch

from oracle import scan_ref


def clouds(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(B, N, 3, generator=g)
    p = p - p.mean(1, keepdim=True)
    return p / p.norm(dim=-1).max(dim=1)[0][:, None, None]


def nerr(got, want):
    """max |got - want| / max(1, max |want|): the normalised error the north-star tolerances are stated on."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return ((got - want).abs().max() / max(1.0, want.abs().max().item())).item()


def oracle_mixers(layers, d, d_state=16):
    refs = []
    for layer in layers:
        r = scan_ref.MambaRef(d, d_state=d_state)
        r.load_state_dict({k: v.detach().cpu() for k, v in layer.mixer.state_dict().items()})
        refs.append(r.eval())
    return refs


def oracle_stack(mixer_model, d, io_dtype=None, taps=None):
    """-> run(x, pos): norm_f(h + residual) after the last layer, or the list of taps after the layers in ``taps``
    (MixerModelForSegmentation.forward, part_segmentation/models/pt_mamba.py:390-416)."""
    refs = oracle_mixers(mixer_model.layers, d)
    r = (lambda t: t) if io_dtype is None else (lambda t: t.to(io_dtype).float())
    ln = torch.nn.functional.layer_norm

    def norm(mod, t):
        return ln(t.float(), (d,), mod.weight.detach().float().cpu(), mod.bias.detach().float().cpu(), mod.eps)

    def run(x, pos):
        # input_ids + pos with torch's own promotion: a bf16 + bf16 add rounds to bf16, anything else is an fp32 add
        h, res, feats = (x.cpu() + pos.cpu()).float(), None, []
        for i, (layer, ref) in enumerate(zip(mixer_model.layers, refs)):
            res = h if res is None else h + res
            h = ref(norm(layer.norm, res), io_dtype=io_dtype)
            if taps is not None and i in taps:
                feats.append(norm(mixer_model.norm_f, h + res))
        return feats if taps is not None else norm(mixer_model.norm_f, h + res)
    return run


def sast_gather_by_order(tokens, pos, order, reverse=True):
    """Token assembly of reference :889-898 + :982-989 from (B,k,G) orders: the k gathers concatenated, then their
    flip -- written out with the reference's own ops."""
    xs, ps = [], []
    for i in range(order.shape[1]):
        idx = order[:, i].unsqueeze(-1)
        xs.append(torch.gather(tokens, 1, idx.expand(-1, -1, tokens.shape[-1])))
        ps.append(torch.gather(pos, 1, idx.expand(-1, -1, pos.shape[-1])))
    x, p = torch.cat(xs, 1), torch.cat(ps, 1)
    if reverse:
        x, p = torch.cat((x, x.flip(1)), 1), torch.cat((p, p.flip(1)), 1)
    return x, p
