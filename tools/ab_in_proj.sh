#!/bin/bash
# A/B of the bf16 step with in_proj through the hand-written kernel (default where its grid fills the chip) and through the
# library GEMM, alternating in one session:  bash tools/ab_in_proj.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for h in None False; do
H=$h python - <<'PY' 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read())['bf16_step']; print(d['value'], 'clouds/s', d['ms_per_step'], 'ms', d['routes'], d['kernels'].get('in_proj_fwd'))"
import os, sys
sys.argv = ["bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-headline"]
from si_mamba_amd import _lib
_lib._hand_in_proj[0] = eval(os.environ["H"])          # None: the default route; False: library GEMM
import bench
bench.main()
PY
done; done
