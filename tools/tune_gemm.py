"""One-off: let PyTorch's TunableOp pick hipBLASLt / rocBLAS solutions for the GEMM shapes of the benchmark
configurations and write them to a CSV (copied to si_mamba_amd/tuned/gemm_gfx950.csv, which
si_mamba_amd.gemm_tuning.enable_tuned_gemms loads read-only).

    python tools/tune_gemm.py gpurun_out/tunableop_results.csv
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
from si_mamba_amd.synthetic import make_clouds

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_results.csv"
dev = torch.device("cuda:0")


def classifier(dtype):
    from si_mamba_amd.point_mamba import PointMamba, default_config
    torch.manual_seed(0)
    m = PointMamba(default_config()).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=0.05, fused=True)
    pts, gt = make_clouds(64, 1024, 0, dev), torch.randint(0, 15, (64,), device=dev)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16")

    def step():
        opt.zero_grad(set_to_none=True)
        with amp:
            loss, _ = m.get_loss_acc(m(pts), gt)
        loss.backward()
        opt.step()
    return step


def mae(dtype):
    from si_mamba_amd.mae import Point_MAE_Mamba, default_mae_config
    torch.manual_seed(0)
    m = Point_MAE_Mamba(default_mae_config()).to(dev).train()
    params = [p for k, p in m.named_parameters() if not k.startswith("decoder_pos_embed.")]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.05, fused=True)
    pts = make_clouds(64, 1024, 0, dev)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16")

    def step():
        opt.zero_grad(set_to_none=True)
        with amp:
            loss = m(pts)
        loss.backward()
        opt.step()
    return step


def seg(dtype):
    from si_mamba_amd.seg import PartSegMamba, get_loss
    torch.manual_seed(0)
    m = PartSegMamba(50).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4, weight_decay=0.05, fused=True)
    pts = make_clouds(16, 2048, 0, dev).transpose(1, 2).contiguous()
    label = torch.nn.functional.one_hot(torch.randint(0, 16, (16,)), 16).float().to(dev)
    target = torch.randint(0, 50, (16, 2048), device=dev)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16")

    def step():
        opt.zero_grad(set_to_none=True)
        with amp:
            o = m(pts, label)
        get_loss()(o.reshape(-1, 50), target.view(-1)).backward()
        opt.step()
    return step


def timeit(step, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tunable.enable(True)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(20)
tunable.set_filename(out)
# fp32 only, and only the workloads named on the command line (default: the judged classifier step).  bf16 is
# deliberately left out: on this image (hipBLASLt 100000-20250912, rocBLAS 5.0.2) trying every bf16
# strided-batched candidate of the classifier step ended in a GPU memory fault ("write access to a read-only
# page") inside one of the candidates -- bf16 runs keep the library defaults.
wanted = sys.argv[2].split(",") if len(sys.argv) > 2 else ["classifier"]
table = {"classifier": classifier, "mae": mae, "seg": seg}
for name, make, dtype in [(w, table[w], "f32") for w in wanted]:
    tunable.tuning_enable(False)
    tunable.enable(False)
    step = make(dtype)
    for _ in range(2):
        step()
    base = timeit(step)
    tunable.enable(True)
    tunable.tuning_enable(True)
    t0 = time.perf_counter()
    step(); torch.cuda.synchronize()
    took = time.perf_counter() - t0
    tunable.tuning_enable(False)
    print(f"{name:10s} {dtype}: default {base:7.2f} ms/step, tuned {timeit(step):7.2f} ms/step (tuning pass {took:.0f} s, "
          f"{len(tunable.get_results())} entries)", flush=True)
    del step
    torch.cuda.empty_cache()
