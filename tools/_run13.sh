set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_mamba.py -m gpu -q --no-header -rf -p no:cacheprovider -x > gpurun_out/tests_conv.log 2>&1; rc=$?
tail -5 gpurun_out/tests_conv.log | cut -c1-250
exit $rc
