"""Where the CPU time of an eager part-segmentation step goes (cProfile over 5 steps; the step is launch-bound at B=16)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd.seg import PartSegMamba, default_seg_config, get_loss
from si_mamba_amd.synthetic import make_clouds

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PartSegMamba(50, default_seg_config(method="HLT")).to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=2e-4, weight_decay=0.05, fused=True)
pts = make_clouds(16, 2048, 0).to(dev).transpose(1, 2).contiguous()
cls_label = torch.nn.functional.one_hot(torch.randint(0, 16, (16,)), 16).float().to(dev)
crit = get_loss()
target = torch.randint(0, 50, (16, 2048), device=dev)
amp = torch.autocast("cuda", dtype=torch.bfloat16)


def step():
    opt.zero_grad(set_to_none=True)
    with amp:
        out = m(pts, cls_label)
    loss = crit(out.reshape(-1, 50), target.view(-1))
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
