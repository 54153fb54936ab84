#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec (fwd+bwd+optimizer) of the PointMamba classifier at
B=64 per GPU, 1024 points -> 128 patches, d=384, 12 blocks (Mamba L = 1024), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      selective-scan forward kernel inside the timed region: algorithmic bytes per launch /
                mean launch duration from events recorded on the launch stream, vs 8 TB/s HBM.
  cpu_baseline  the oracle (CPU restatement) running the same step on a bounded sample (two steps of 4 clouds after a
                warm-up step) on the host cores.
plus "headline_scan" (the north-star micro-shape B=256, L=128, D=768, N=16, measured after the timed region, fp32 and
bf16 I/O), "kernels" (mean ms per launch of the two scan kernels, event-bracketed inside the timed region) and
"kernels_outside_timed_region" (the other HIP kernels, from two more steps after it).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def scan_fwd_bytes(B, D, L, N, s=4):
    """u, delta, z, out + B, C + A + D, bias."""
    return 4 * B * D * L * s + 2 * B * N * L * s + 4 * D * N + 8 * D


def scan_bwd_bytes(B, D, L, N, s=4):
    return 7 * B * D * L * s + 2 * B * N * L * (s + 4) + 8 * D * N + 16 * D


SCAN_SOURCES = ("common.h", "scan_common.h", "scan_xlane.h", "scan_fwd.hip", "scan_fwd_seq.hip", "scan_bwd.hip",
                "scan_bwd_seq.hip")


def scan_sources_sha256():
    """Hash of the scan kernels' sources: profiles/traffic.json records the one its PMC passes were taken on."""
    import hashlib
    h = hashlib.sha256()
    for name in SCAN_SOURCES:
        with open(os.path.join(ROOT, "si_mamba_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def traffic_from_profiles(kernel, shape):
    """(HBM bytes per launch, provenance) from the committed rocprofv3 PMC summary profiles/traffic.json -- a constant
    measured by tools/pmc_traffic.sh, NOT measured in this run; (None, why) when no entry matches or when the scan
    sources have changed since those passes were taken (the number would be stale)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None, "profiles/traffic.json unreadable"
    val = rec.get(f"{kernel}:{'x'.join(map(str, shape))}")
    if val is None:
        return None, "no PMC entry for this kernel and shape in profiles/traffic.json"
    if rec.get("_sources_sha256") != scan_sources_sha256():
        return None, "profiles/traffic.json was measured on other scan sources (stale): re-run tools/pmc_traffic.sh"
    return val, f"profiles/traffic.json ({rec.get('_measured', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}); " \
                "a committed constant, not measured in this run"


def cpu_baseline(npts, groups, seed=0):
    """Oracle port of the same training step (same architecture, fp32) on a bounded sample: one warm-up step of 2
    clouds (thread pools, allocator), then two timed steps of 4 clouds each (about 20 s of CPU work)."""
    from oracle import fps_ref, scan_ref, spectral_ref
    from si_mamba_amd.point_mamba import PointMamba, default_config
    from si_mamba_amd.synthetic import make_clouds
    # the oracle's per-timestep torch ops are tiny: more threads only add fork/join latency
    torch.set_num_threads(min(16, torch.get_num_threads()))
    torch.manual_seed(0)
    cfg = default_config(num_group=groups, drop_path=0.)
    m = PointMamba(cfg)
    for layer in m.blocks.layers:                          # CPU mixers: the oracle restatement
        ref = scan_ref.MambaRef(cfg.trans_dim, layer_idx=layer.layer_idx)
        ref.load_state_dict(layer.mixer.state_dict())
        layer.mixer = ref

    def spectral_order(center):
        adj = spectral_ref.create_graph_from_feature_space(center, cfg.knn_graph, cfg.alpha, cfg.symmetric,
                                                           cfg.self_loop, cfg.binary)
        _, vecs, _, _ = spectral_ref.calc_top_k_eigenvalues_eigenvectors(adj, cfg.k_top_eigenvectors, True)
        return spectral_ref.spectral_orders(vecs)

    m.spectral_order = spectral_order
    m.group_divider.fps_fn = fps_ref.sample_farthest_points
    m.train()
    def one_step(B, seed):
        pts = make_clouds(B, npts, seed, "cpu")
        gt = torch.randint(0, cfg.cls_dim, (B,))
        t0 = time.perf_counter()
        loss, _ = m.get_loss_acc(m(pts), gt)
        loss.backward()
        m.zero_grad(set_to_none=True)
        return time.perf_counter() - t0

    one_step(2, seed)
    B, steps = 4, 2
    dts = [one_step(B, seed + 1 + i) for i in range(steps)]
    dt = sum(dts)
    return {"value": round(B * steps / dt, 4), "unit": "point-clouds/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} steps of {B} clouds after a 2-cloud warm-up step, same 12-block d=384 model, fwd+bwd, "
                      f"oracle mixers + torch.linalg.eigh ordering ({dt:.1f} s; per step "
                      f"{', '.join(f'{x:.1f}' for x in dts)} s)"}


def headline_scan(device, iters=30):
    """North-star micro-shape: scan fwd (and bwd) at (B,D,L,N) = (256,768,128,16), fp32.

    Two clocks per direction: ``ms`` brackets ONE call of the Python op with events (what round 1 reported: it
    includes the ~25 us launch gap of an idle stream), ``kernel_ms`` divides an event-bracketed train of
    back-to-back raw C-ABI launches by their number -- the kernel's own duration plus the ~1.5 us boundary, the
    figure rocprofv3 --kernel-trace reports (profiles/)."""
    from si_mamba_amd import _lib, selective_scan_fn
    from si_mamba_amd.synthetic import scan_inputs
    B, D, L, N = 256, 768, 128, 16
    t = {k: (v.to(device) if v is not None else None) for k, v in scan_inputs(B, D, L, N, seed=0).items()}
    leaves = [t[k].requires_grad_(True) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")]
    res = {}
    for mode in ("fwd", "bwd"):
        times = []
        for i in range(iters + 5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if mode == "fwd":
                with torch.no_grad():
                    a.record()
                    selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["z"], t["delta_bias"], True)
                    b.record()
            else:
                out = selective_scan_fn(*leaves[:6], z=leaves[6], delta_bias=leaves[7], delta_softplus=True)
                a.record()
                torch.autograd.grad(out, leaves, t["dout"])
                b.record()
            torch.cuda.synchronize()
            if i >= 5:
                times.append(a.elapsed_time(b))
        times.sort()
        med = times[len(times) // 2]
        nbytes = scan_fwd_bytes(B, D, L, N) if mode == "fwd" else scan_bwd_bytes(B, D, L, N)
        res[mode] = {"ms": round(med, 4), "GB/s": round(nbytes / med / 1e6, 1),
                     "frac_of_8TBs": round(nbytes / med / 1e6 / HBM_PEAK_GBS, 4)}
    # back-to-back raw launches through the C ABI
    lib = _lib.load()
    st = _lib.stream_ptr(device)
    d = {k: v.detach() for k, v in t.items() if v is not None}
    out = torch.empty_like(d["u"])
    du, dd, dz = (torch.empty_like(d["u"]) for _ in range(3))
    acc = _lib.scan_bwd_accumulators(B, D, L, N, True, True, device)
    # the forward is timed without checkpoints (the north-star figure: what an inference call runs); the backward
    # reads the checkpoints of one untimed forward in the layout the library picks for this shape
    ck_step = lib.simamba_scan_ckpt_step(B, D, L, N, 0)
    nck = lib.simamba_scan_ckpt_floats(B, D, L, N, ck_step)
    ck = torch.empty(nck, device=device) if nck else None
    ck16 = None

    def fwd(ckpt=None):
        return lib.simamba_selective_scan_fwd(d["u"].data_ptr(), d["delta"].data_ptr(), d["A"].data_ptr(),
                                              d["B"].data_ptr(), d["C"].data_ptr(), d["D"].data_ptr(), d["z"].data_ptr(),
                                              d["delta_bias"].data_ptr(), out.data_ptr(), _lib.ptr(ckpt), None,
                                              B, D, L, N, 0, 1, 0, 0, 0, 0, ck_step, 0, st)

    def bwd():
        return lib.simamba_selective_scan_bwd(d["u"].data_ptr(), d["delta"].data_ptr(), d["A"].data_ptr(),
                                              d["B"].data_ptr(), d["C"].data_ptr(), d["D"].data_ptr(), d["z"].data_ptr(),
                                              d["delta_bias"].data_ptr(), d["dout"].data_ptr(), _lib.ptr(ck),
                                              du.data_ptr(), dd.data_ptr(), acc[0].data_ptr(), acc[1].data_ptr(),
                                              acc[2].data_ptr(), acc[3].data_ptr(), dz.data_ptr(), acc[4].data_ptr(),
                                              B, D, L, N, 0, 1, 0, 0, 0, 0, 0, ck_step, st)

    assert fwd(ck) == 0

    for mode, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(5):
            assert fn() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        nbytes = scan_fwd_bytes(B, D, L, N) if mode == "fwd" else scan_bwd_bytes(B, D, L, N)
        res[mode].update({"kernel_ms": round(ms, 4), "kernel_GB/s": round(nbytes / ms / 1e6, 1),
                          "kernel_frac_of_8TBs": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)})
    # the same trains with bf16 operands and results (fp32 A / D / bias / accumulators, fp32 arithmetic inside): the I/O
    # type of the autocast configurations; half the bytes through the same instruction stream
    h = {k: d[k].to(torch.bfloat16) for k in ("u", "delta", "z", "B", "C", "dout")}
    outh = torch.empty_like(h["u"])
    duh, ddh, dzh = (torch.empty_like(h["u"]) for _ in range(3))

    ck_step16 = lib.simamba_scan_ckpt_step(B, D, L, N, _lib.BF16)
    nck16 = lib.simamba_scan_ckpt_floats(B, D, L, N, ck_step16)
    ck16 = torch.empty(nck16, device=device) if nck16 else None

    def fwd16(ckpt=None):
        return lib.simamba_selective_scan_fwd(h["u"].data_ptr(), h["delta"].data_ptr(), d["A"].data_ptr(),
                                              h["B"].data_ptr(), h["C"].data_ptr(), d["D"].data_ptr(), h["z"].data_ptr(),
                                              d["delta_bias"].data_ptr(), outh.data_ptr(), _lib.ptr(ckpt), None,
                                              B, D, L, N, _lib.BF16, 1, 0, 0, 0, 0, ck_step16, 0, st)

    def bwd16():
        return lib.simamba_selective_scan_bwd(h["u"].data_ptr(), h["delta"].data_ptr(), d["A"].data_ptr(),
                                              h["B"].data_ptr(), h["C"].data_ptr(), d["D"].data_ptr(), h["z"].data_ptr(),
                                              d["delta_bias"].data_ptr(), h["dout"].data_ptr(), _lib.ptr(ck16),
                                              duh.data_ptr(), ddh.data_ptr(), acc[0].data_ptr(), acc[1].data_ptr(),
                                              acc[2].data_ptr(), acc[3].data_ptr(), dzh.data_ptr(), acc[4].data_ptr(),
                                              B, D, L, N, _lib.BF16, 1, 0, 0, 0, 0, 0, ck_step16, st)

    assert fwd16(ck16) == 0

    res["bf16_io"] = {}
    for mode, fn in (("fwd", fwd16), ("bwd", bwd16)):
        for _ in range(5):
            assert fn() == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        nbytes = scan_fwd_bytes(B, D, L, N, 2) if mode == "fwd" else scan_bwd_bytes(B, D, L, N, 2)
        res["bf16_io"][mode] = {"kernel_ms": round(ms, 4), "kernel_GB/s": round(nbytes / ms / 1e6, 1),
                                "kernel_frac_of_8TBs": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    res["shape"] = [B, D, L, N]
    res["note"] = ("ms: one event-bracketed call of the Python op (includes the launch gap of an idle stream); kernel_ms: "
                   "a train of back-to-back C-ABI launches / their number (kernel + boundary; bwd includes its memset "
                   "node); fwd kernel: scan_fwd_seq_kernel<float,true,2> (two lanes per channel, 32-step chunks)")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU")
    ap.add_argument("--npoints", type=int, default=1024)
    ap.add_argument("--groups", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--scan-variant", type=int, default=0,
                    help="A/B tool: force one forward-scan kernel (include/simamba.h SIMAMBA_SCAN_*; 0 = the library's "
                         "choice, which is what every reported line uses)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-headline", action="store_true")
    ap.add_argument("--no-bf16-step", action="store_true",
                    help="skip the short bf16-autocast measurement that follows the timed fp32 region")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-tuned-gemm", action="store_true",
                    help="library-default GEMM solutions instead of si_mamba_amd/tuned/gemm_gfx950.csv")
    args = ap.parse_args()

    from si_mamba_amd import _lib
    from si_mamba_amd import dist as sdist
    from si_mamba_amd.gemm_tuning import enable_tuned_gemms
    from si_mamba_amd.point_mamba import PointMamba, default_config
    from si_mamba_amd.synthetic import make_clouds
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the product path has no CPU fallback")
    # stdout carries exactly one JSON line: RCCL prints a version banner to fd 1 when a communicator is created,
    # so everything else this process (and the libraries it loads) writes to stdout goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # RCCL ("nccl") over xGMI; SIMAMBA_DIST_BACKEND=gloo only to rehearse the N > 1 code path on a one-GPU box
    rank, world = sdist.init_dist(os.environ.get("SIMAMBA_DIST_BACKEND", "nccl") if args.gpus > 1 else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    _lib.load()
    # library GEMMs take the solutions recorded for this step's shapes (si_mamba_amd/gemm_tuning.py: every fp32 GEMM
    # of the step, and the plain -- not strided-batched -- bf16 ones); look-up only, nothing is tuned inside the run
    tuned = (not args.no_tuned_gemm) and enable_tuned_gemms()

    torch.manual_seed(0)                                    # same init on every rank
    cfg = default_config(num_group=args.groups)             # cfgs/finetune_scan_hardest.yaml model block
    model = PointMamba(cfg).to(device).train()
    ddp = sdist.wrap_ddp(model, device)
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.05, fused=True)
    pts = make_clouds(args.batch, args.npoints, seed=rank, device=device)
    gt = torch.randint(0, cfg.cls_dim, (args.batch,), generator=torch.Generator().manual_seed(rank)).to(device)
    amp = torch.autocast("cuda", dtype=torch.bfloat16, enabled=(args.dtype == "bf16"))

    def step(amp=amp):
        opt.zero_grad(set_to_none=True)
        with amp:
            logits = ddp(pts)
            loss, _ = model.get_loss_acc(logits, gt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)     # grad_norm_clip: 10
        opt.step()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # events around the scan launches only (the roofline kernels): a pair of event records around every launch of the
    # step costs 0.5 ms (fp32) to 7 ms (bf16, CPU-bound then) of the step it measures
    _lib.enable_kernel_timing(not args.no_kernel_timing, only=("scan_fwd", "scan_bwd"))
    _lib._scan_variant[0] = args.scan_variant
    gc.collect()        # a full collection of the interpreter (~80 ms with this many live tensors) now, not inside the region
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = te.item()
    ktimes = _lib.kernel_times()
    other = {}
    if not args.no_kernel_timing:
        # the other HIP kernels' mean durations, for the record: two more steps with every launch bracketed, untimed
        _lib.enable_kernel_timing(True)
        for _ in range(2):
            step()
        sync()
        other = {k: v for k, v in _lib.kernel_times().items() if k not in ktimes}
    _lib.enable_kernel_timing(False)
    assert torch.isfinite(loss).item(), "non-finite loss in the timed region"

    # The same step under bf16 autocast (what the reference's pre-training and segmentation runners use,
    # tools/runner_pretrain.py:243), measured AFTER the timed region so that the driver's record carries it: not the
    # headline, same model / data / optimizer state continued, 3 warm-up + 10 timed steps, max over ranks.
    bf16_step = None
    if args.dtype == "f32" and not args.no_bf16_step:
        amp16 = torch.autocast("cuda", dtype=torch.bfloat16)
        for _ in range(3):
            step(amp16)
        gc.collect()
        sync()
        t1 = time.perf_counter()
        for _ in range(10):
            loss16 = step(amp16)
        sync()
        el16 = time.perf_counter() - t1
        if world > 1:
            te = torch.tensor([el16], device=device, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el16 = te.item()
        bf16_step = {"value": round(args.batch * world * 10 / el16, 2), "unit": "point-clouds/s",
                     "ms_per_step": round(1e2 * el16, 3), "steps": 10, "warmup": 3,
                     "dtype": "bf16 autocast (parameters, optimizer, reductions, scan state fp32)",
                     "finite_loss": bool(torch.isfinite(loss16).item()),
                     "note": "measured after the timed fp32 region on the same model and data; not the headline"}
        if not args.no_kernel_timing:
            # two more bf16 steps with every HIP launch event-bracketed (untimed), and which routes the mixer took:
            # delta formed inside the scans (scan_dt_fwd) and out_proj + add + LayerNorm as one kernel (out_proj_add_ln)
            _lib.counters.clear()
            _lib.enable_kernel_timing(True)
            for _ in range(2):
                step(amp16)
            sync()
            bf16_step["kernels"] = {k: {"launches": v[0], "mean_ms": round(v[1], 4)}
                                    for k, v in _lib.kernel_times().items()}
            bf16_step["routes"] = dict(_lib.counters)
            _lib.enable_kernel_timing(False)

    if rank == 0:
        L = 2 * cfg.k_top_eigenvectors * args.groups
        D, N = 2 * cfg.trans_dim, 16
        s = 4 if args.dtype == "f32" else 2
        clouds = args.batch * world * args.steps
        out = {
            "metric": "point-clouds/sec (fwd+bwd)", "value": round(clouds / elapsed, 2), "unit": "point-clouds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"PointMamba classifier train step (FPS+kNN grouping, encoder, spectral SAST "
                                   f"ordering, 12 Mamba blocks d=384 at L={L}, head, AdamW), "
                                   f"{args.npoints} pts -> {args.groups} patches",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch,
                       "npoints": args.npoints, "patches": args.groups, "mamba_seq_len": L,
                       "parallelism": f"dp{world}",
                       "dist_backend": (dist.get_backend() if world > 1 else None),
                       "communicator_ranks": (dist.get_world_size() if world > 1 else 1),
                       "library_gemm": "tuned solution table (si_mamba_amd/tuned/gemm_gfx950.csv)" if tuned
                       else "library defaults"},
        }
        if "scan_fwd" in ktimes:
            n, ms = ktimes["scan_fwd"]
            nbytes = scan_fwd_bytes(args.batch, D, L, N, s)
            ach = nbytes / ms / 1e6
            rows = args.batch * D
            tname = "float" if s == 4 else "bf16"
            auto = _lib.load().simamba_scan_fwd_auto_variant(args.batch, D)        # the library's own choice
            # the in-step forward also writes the backward's checkpoints ("train" entries of traffic.json)
            tr_f = traffic_from_profiles("scan_fwd_train", (args.batch, D, L, N)) if s == 4 else (None, "fp32 only")
            fwd_kernel = {_lib.SCAN_MIX: f"scan_fwd_seq_mix_kernel<{tname},true>",
                          _lib.SCAN_LPC2: f"scan_fwd_seq_kernel<{tname},true,2>",
                          _lib.SCAN_LPC4: f"scan_fwd_seq_kernel<{tname},true,4>"}.get(
                              auto, f"scan_fwd_kernel<{tname},{16 if L >= 768 else 8}>")
            out["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4),
                               "traffic": tr_f[0], "traffic_source": tr_f[1],
                               "kernel": fwd_kernel,
                               "kernel_source": "derived from simamba_scan_fwd_auto_variant(batch, dim), not observed",
                               "shape_BDLN": [args.batch, D, L, N], "algorithmic_bytes": nbytes,
                               "launches": n, "mean_ms": round(ms, 4),
                               "note": "VALU-bound on CDNA4, not HBM-bound (DESIGN.md 4.1): 5 VALU per (row, step, "
                                       "state) incl. one quarter-rate v_exp_f32 and two DPP operands; PMC: VALU "
                                       ">90 % busy, HBM traffic 1.0x algorithmic. At this shape 1536 waves on 1024 "
                                       "SIMDs: half of the SIMDs carry two waves (the mixed 2 + 4 lanes launch that "
                                       "evens them out is 14 % faster alone, 15 % slower behind the 400 MB the "
                                       "preceding kernel has just written: DESIGN.md 4.1)"}
            out["kernels"] = {k: {"launches": v[0], "mean_ms": round(v[1], 4)} for k, v in ktimes.items()}
            out["kernels_outside_timed_region"] = {k: {"launches": v[0], "mean_ms": round(v[1], 4)}
                                                   for k, v in other.items()}
            if "scan_bwd" in ktimes:
                bb = scan_bwd_bytes(args.batch, D, L, N, s)
                nb, msb = ktimes["scan_bwd"]
                out["kernels"]["scan_bwd"]["GB/s"] = round(bb / msb / 1e6, 1)
                # the backward is the larger HIP kernel of the step by time: same accounting, for completeness
                ck = _lib.load().simamba_scan_ckpt_step(args.batch, D, L, N, _lib.F32 if s == 4 else _lib.BF16)
                seq = ck == _lib.CKPT_SEQ
                tr_b = traffic_from_profiles("scan_bwd", (args.batch, D, L, N)) if s == 4 else (None, "fp32 only")
                out["roofline_scan_bwd"] = {
                    "bound": "hbm", "achieved": round(bb / msb / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(bb / msb / 1e6 / HBM_PEAK_GBS, 4),
                    "traffic": tr_b[0], "traffic_source": tr_b[1],
                    "kernel": (f"scan_bwd_seq_kernel<{tname},true>" if seq else f"scan_bwd_kernel<{tname},8>"),
                    "algorithmic_bytes": bb, "launches": nb, "mean_ms": round(msb, 4),
                    "note": ("sequential lanes-per-channel backward (csrc/scan_bwd_seq.hip): VALU-bound, PMC: ~26 VALU "
                             "per (row, step, state) at 91 % VALU-busy, 3 waves/SIMD; HBM traffic 1.24x the "
                             "algorithmic bytes (16-step state checkpoints + dB/dC flush atomics); DESIGN.md 4.2"
                             if seq else
                             "row-scan backward: VALU-bound (27.6 VALU per (row, step, state), 76 % busy at 2 "
                             "waves/SIMD); DESIGN.md 4.2")}
        if bf16_step is not None:
            out["bf16_step"] = bf16_step
        if world == 1 and not args.no_headline:
            del opt
            torch.cuda.empty_cache()
            out["headline_scan"] = headline_scan(device)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.npoints, args.groups)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
