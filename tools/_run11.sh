set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_out_norm.py tests/test_gpu_model.py tests/test_gpu_mamba.py -m gpu -q --no-header -rf -p no:cacheprovider -x > gpurun_out/tests_on.log 2>&1; rc=$?
tail -25 gpurun_out/tests_on.log | cut -c1-250
exit $rc
