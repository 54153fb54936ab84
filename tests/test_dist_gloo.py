"""CPU, world_size 2 over gloo: the data-parallel plumbing (si_mamba_amd.dist) and the gradient
equivalence "one process with the whole batch == two ranks with half each + DDP all-reduce".

The HIP mixers cannot run on CPU ranks, so the model here uses the oracle mixer (tests may import
oracle/); what is under test is the sharding, the collectives and the DDP wrap that bench.py uses.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import scan_ref
from si_mamba_amd import dist as sdist


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm = nn.LayerNorm(32)
        self.mixer = scan_ref.MambaRef(32, d_state=4)
        self.head = nn.Linear(32, 5)

    def forward(self, x):
        return self.head(self.mixer(self.norm(x)).mean(1))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, w = sdist.init_dist("gloo")
    assert (r, w) == (rank, world) and sdist.get_dist_info() == (rank, world)
    # reduce_tensor = all-reduce(SUM)/world ; gather_tensor = all-gather + cat  (reference utils/dist_utils.py)
    red = sdist.reduce_tensor(torch.tensor([float(rank + 1)]))
    gat = sdist.gather_tensor(torch.tensor([rank, rank + 10]))
    assert red.item() == pytest.approx(1.5) and gat.tolist() == [0, 10, 1, 11]
    torch.manual_seed(0)
    model = Tiny()
    ddp = sdist.wrap_ddp(model, torch.device("cpu"))
    g = torch.Generator().manual_seed(123)
    x, y = torch.randn(4, 12, 32, generator=g), torch.randint(0, 5, (4,), generator=g)
    xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]       # shard by sample
    loss = nn.functional.cross_entropy(ddp(xs), ys)
    loss.backward()
    if rank == 0:
        ret["grads"] = {k: p.grad.clone() for k, p in model.named_parameters()}
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_match_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    torch.manual_seed(0)
    model = Tiny()
    g = torch.Generator().manual_seed(123)
    x, y = torch.randn(4, 12, 32, generator=g), torch.randint(0, 5, (4,), generator=g)
    nn.functional.cross_entropy(model(x), y).backward()
    for k, p in model.named_parameters():
        torch.testing.assert_close(ret["grads"][k], p.grad, rtol=1e-4, atol=1e-6)


def test_single_process_helpers_are_identity():
    assert sdist.get_dist_info() == (0, 1)
    t = torch.arange(3.0)
    assert torch.equal(sdist.reduce_tensor(t), t) and torch.equal(sdist.gather_tensor(t), t)
    m = nn.Linear(2, 2)
    assert sdist.wrap_ddp(m) is m          # no process group: left unwrapped
