"""Diagnostic: which aten ops copy activation-sized tensors in one Mamba block fwd+bwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from si_mamba_amd.block import create_block
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = create_block(384, layer_idx=0, drop_path=0.0).to(dev)
x = torch.randn(64, 1024, 384, device=dev, requires_grad=True)
res = torch.randn(64, 1024, 384, device=dev, requires_grad=True)
for _ in range(2):
    h, r = blk(x, res); (h.sum() + r.sum()).backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    h, r = blk(x, res); (h.sum() + r.sum()).backward()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=90))
