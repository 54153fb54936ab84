set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_scan_dt.py tests/test_gpu_mamba.py -m gpu -q --no-header -rf -p no:cacheprovider -x > gpurun_out/tests_scan.log 2>&1 || { tail -20 gpurun_out/tests_scan.log; exit 1; }
tail -1 gpurun_out/tests_scan.log
timeout -k 10 300 python tools/bench_scan.py --modes fwd --iters 30 --shapes 256x768x128,64x768x1024 2>&1 | grep -v Warn | head -4
