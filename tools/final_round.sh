#!/bin/bash
# End-of-round GPU session: parity tests, the bench line, rocprofv3 kernel stats of the fp32 and the bf16 step, MFMA counters.
# Every step is bounded; the session stops at the first step that fails, times out or is killed.
#   usage (on the GPU box): bash tools/final_round.sh            -> gpurun_out/final/
set -u
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 "$to" "$@" > "$O/$name.log" 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -ne 0 ]; then echo "[$name] failed / timed out: stopping"; tail -5 "$O/$name.log"; exit 1; fi
}
run tests 900 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider
tail -2 $O/tests.log
run bench 600 python bench.py --steps 10 --warmup 3
grep '^{' $O/bench.log | tail -1 > $O/bench.json
run prof_f32 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 bench.py --steps 7 --warmup 1 --no-cpu-baseline --no-headline --no-kernel-timing --no-bf16-step
cp "$(find $O/prof_f32 -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv
run prof_bf16 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 bench.py --dtype bf16 --steps 7 --warmup 1 --no-cpu-baseline --no-headline --no-kernel-timing --no-bf16-step
cp "$(find $O/prof_bf16 -name '*kernel_stats.csv' | head -1)" $O/bench_bf16_kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
DTYPE=bf16 bash tools/pmc_bench.sh $O/pmc_bf16 > $O/mfma_counters_bench_bf16.txt 2>&1 || { echo "[pmc bf16] failed"; exit 1; }
DTYPE=f32 bash tools/pmc_bench.sh $O/pmc_f32 > $O/mfma_counters_bench_f32.txt 2>&1 || { echo "[pmc f32] failed"; exit 1; }
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
tail -1 $O/bench.json | cut -c1-300
