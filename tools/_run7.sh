set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 6 0 6; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline --no-bf16-step --scan-variant $v > gpurun_out/bench_v$v.log 2> gpurun_out/bench_v$v.err; tail -1 gpurun_out/bench_v$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['ms_per_step'], d['kernels'])"
done
