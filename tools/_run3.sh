set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-headline > gpurun_out/prof.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r03a_bench_kernel_stats.csv; head -30 $f | cut -c1-150
find gpurun_out/prof -name "*.csv" -size +1M -delete
tail -1 gpurun_out/prof.log | cut -c1-200
