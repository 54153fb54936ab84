set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_scan_dt.py -m gpu -q --no-header -rf -p no:cacheprovider -x > gpurun_out/tests_dt.log 2>&1 || { tail -20 gpurun_out/tests_dt.log; exit 1; }
tail -1 gpurun_out/tests_dt.log
timeout -k 10 300 python tools/bench_dt_fusion.py 64 > gpurun_out/bench_dt_fusion.txt 2>&1; cat gpurun_out/bench_dt_fusion.txt | tail -8
