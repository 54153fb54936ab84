"""bench.py's main() with the bf16 mixers forced back onto conv kernel + two library GEMMs (A/B of csrc/xdt_proj_bf16.hip):
   python tools/bench_no_bf16_xdt.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-headline"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from si_mamba_amd import mamba_inner
_real = mamba_inner.xdt_proj_fused_ok
mamba_inner.xdt_proj_fused_ok = lambda x, *a, **k: x.dtype == torch.float32 and _real(x, *a, **k)
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
bench.main()
