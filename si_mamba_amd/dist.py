"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

Mirrors reference utils/dist_utils.py:9-54 (init_dist / reduce_tensor / gather_tensor) and the DDP
wrap of tools/runner_finetune.py:124-125.  The hot-path kernels have no cross-sample term, so the
only collective per step is the bucketed gradient all-reduce DDP issues on its own RCCL stream.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_dist(backend: str | None = None):
    """Initialise from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and "MASTER_ADDR" not in os.environ:
        return 0, 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", str(rank % max(torch.cuda.device_count(), 1))))
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def reduce_tensor(tensor, world_size=None):
    """all-reduce(SUM) / world -- reference utils/dist_utils.py:41-48."""
    _, ws = get_dist_info()
    world_size = ws if world_size is None else world_size
    if ws == 1:
        return tensor.clone()
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    return rt / world_size


def gather_tensor(tensor, world_size=None):
    """all-gather + cat on dim 0 -- reference utils/dist_utils.py:50-54."""
    _, ws = get_dist_info()
    if ws == 1:
        return tensor.clone()
    outs = [torch.empty_like(tensor) for _ in range(ws)]
    dist.all_gather(outs, tensor)
    return torch.cat(outs, dim=0)


def wrap_ddp(model, device=None, bucket_cap_mb: int = 64):
    """DDP tuned for xGMI: the 49 MB of fp32 gradients go out as ONE bucket (ring all-reduce is
    per-link bound, fewer/larger messages win), gradients alias the bucket, static graph."""
    if not (dist.is_available() and dist.is_initialized()):
        return model                      # plain single-process run: nothing to synchronise
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = None if device is None or device.type != "cuda" else [device.index]
    # broadcast_buffers=True is DDP's default and what the reference gets (tools/runner_finetune.py:124-125):
    # BatchNorm running statistics follow rank 0 (a few KB per step)
    return DDP(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
               static_graph=True, broadcast_buffers=True)
