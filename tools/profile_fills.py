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
# kernel-side view: for every fill kernel, the innermost CPU op whose time range covers the launch (correlated by the
# launch call's timestamp when the profiler lost the link)
ev = list(prof.events())
cpu = sorted([e for e in ev if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: e.time_range.start)
import bisect
starts = [e.time_range.start for e in cpu]
cnt = collections.Counter()
launches = [e for e in cpu if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel", "hipMemsetAsync", "hipMemcpyAsync")]
print("launch calls:", collections.Counter(e.name for e in launches))
def innermost(t, tid):
    best = None
    i = bisect.bisect_right(starts, t)
    for e in reversed(cpu[max(0, i - 400):i]):
        if e.thread == tid and e.time_range.start <= t <= e.time_range.end and not e.name.startswith("hip"):
            if best is None or (e.time_range.end - e.time_range.start) < (best.time_range.end - best.time_range.start):
                best = e
    return best
for l in launches:
    if l.name == "hipMemsetAsync" or any("FillFunctor" in k.name for k in l.kernels):
        par = innermost(l.time_range.start, l.thread)
        cnt[(l.name, par.name if par else "?", str(par.input_shapes)[:60] if par else "")] += 1
for k, v in cnt.most_common(30):
    print(v, k)
