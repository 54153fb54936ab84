set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_bf16 && mkdir -p gpurun_out/prof_bf16
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bf16 -o bf16 --output-format csv -- python3 bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-headline --no-kernel-timing > gpurun_out/prof_bf16/bench.log 2>&1; rc=$?
find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_bf16_kernel_stats.csv
find gpurun_out/prof_bf16 -name "*_kernel_trace.csv" -delete
find gpurun_out/prof_bf16 -name "*agent_info.csv" -delete
tail -2 gpurun_out/prof_bf16/bench.log | cut -c1-400
exit $rc
