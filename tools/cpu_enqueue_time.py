"""CPU time to enqueue one training step against the GPU time it takes (is the eager step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from si_mamba_amd import _lib
from si_mamba_amd.point_mamba import PointMamba, default_config
from si_mamba_amd.synthetic import make_clouds
from si_mamba_amd.gemm_tuning import enable_tuned_gemms
dev = torch.device("cuda:0")
enable_tuned_gemms()
for dtype in ("f32", "bf16"):
    cfg = default_config(num_group=128)
    model = PointMamba(cfg).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.05, fused=True)
    pts = make_clouds(64, 1024, seed=0, device=dev)
    gt = torch.randint(0, cfg.cls_dim, (64,)).to(dev)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=(dtype == "bf16"))
    def step():
        opt.zero_grad(set_to_none=True)
        with amp:
            loss, _ = model.get_loss_acc(model(pts), gt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
    for timing in (False, True):
        _lib.enable_kernel_timing(timing)
        for _ in range(3): step()
        torch.cuda.synchronize()
        enq = []
        t0 = time.perf_counter()
        for _ in range(10):
            a = time.perf_counter(); step(); enq.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{dtype} kernel-timing={timing}: enqueue {1e3 * (t1 - t0) / 10:.1f} ms/step (min {1e3 * min(enq):.1f}), "
              f"wall {1e3 * (t2 - t0) / 10:.1f} ms/step", flush=True)
    _lib.enable_kernel_timing(False)
