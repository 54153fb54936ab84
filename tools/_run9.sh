set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-step > gpurun_out/bench4.log 2> gpurun_out/bench4.err; tail -1 gpurun_out/bench4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels']); h=d['headline_scan']; print(h['fwd'], h['bwd'], h['bf16_io'])"
