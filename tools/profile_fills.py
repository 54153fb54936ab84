"""Which ops of the training step launch fill / memset kernels (tuning tool)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from si_mamba_amd.point_mamba import PointMamba, default_config
from si_mamba_amd.synthetic import make_clouds
from si_mamba_amd.gemm_tuning import enable_tuned_gemms
dev = torch.device("cuda:0")
enable_tuned_gemms()
cfg = default_config(num_group=128)
model = PointMamba(cfg).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.05, fused=True)
pts = make_clouds(64, 1024, seed=0, device=dev)
gt = torch.randint(0, cfg.cls_dim, (64,)).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model.get_loss_acc(model(pts), gt)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
ev = [e for e in prof.events()]
cpu_ops = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
for e in ev:
    n = e.name
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    # top-level-ish ops that have kernels attached
    for k in e.kernels:
        kn = k.name
        if "FillFunctor" in kn or "fillBuffer" in kn or "copyBuffer" in kn or "direct_copy" in kn or "neg_kernel" in kn or "exp_kernel" in kn:
            cnt[(kn[:60], n, str(e.input_shapes)[:70])] += 1
for k, v in cnt.most_common(45):
    print(v, k)
