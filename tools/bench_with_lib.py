"""Runs bench.py's main() against an alternative build of the library (A/B tuning tool):
   python tools/bench_with_lib.py tools/alt/libsimamba_x.so --steps 10 --warmup 3 --no-cpu-baseline --no-headline"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from si_mamba_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
