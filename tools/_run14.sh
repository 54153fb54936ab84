set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_out_norm.py tests/test_gpu_model.py tests/test_gpu_mamba.py tests/test_gpu_mae.py tests/test_gpu_seg.py tests/test_gpu_pinned.py -m gpu -q --no-header -rf -p no:cacheprovider -x > gpurun_out/tests_sub.log 2>&1; rc=$?
tail -4 gpurun_out/tests_sub.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_sub.json 2>gpurun_out/bench_sub.err; rc=$?
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/bench_sub.json') if x.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['bf16_step']['value'], d['bf16_step']['ms_per_step'], d['roofline']['frac'], d['roofline_scan_bwd']['frac'])
PY
exit $rc
