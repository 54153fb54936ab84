set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_mae.py tests/test_gpu_seg.py tests/test_gpu_model.py -m gpu -q --no-header -rf -p no:cacheprovider -x -k "config4 or config5 or distinct or precomputed" > gpurun_out/tests2.log 2>&1; rc=$?
tail -15 gpurun_out/tests2.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-headline > gpurun_out/bench2.log 2> gpurun_out/bench2.err; tail -1 gpurun_out/bench2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('bf16_step'), d['roofline']['traffic_source'])"
