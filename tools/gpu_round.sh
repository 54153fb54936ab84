#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats.  Stops after any timeout/kill.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] timed out/killed: stopping"; exit 1; fi
  return $rc
}
run tests 900 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider
tail -4 gpurun_out/tests.log
run bench 600 python bench.py --steps 5 --warmup 2
tail -1 gpurun_out/bench.log
if [ "${PROFILE:-1}" = "1" ]; then
  run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline
  find gpurun_out/prof -name "*kernel_stats.csv" | head -3
fi
