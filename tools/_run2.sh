set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider -x > gpurun_out/tests.log 2>&1; rc=$?
tail -5 gpurun_out/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -1 gpurun_out/bench.log
