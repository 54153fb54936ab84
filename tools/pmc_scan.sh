#!/bin/bash
# PMC passes (separate runs, no tracing besides --kernel-trace) for the scan kernels at the micro-shape.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
SH=${SHAPES:-256x768x128}
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc/p$i -- python tools/bench_scan.py --shapes $SH --iters 3 --modes ${MODES:-fwd,bwd} > gpurun_out/pmc/p$i.log 2>&1
  rc=$?; echo "[pmc pass $i] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "simamba" not in k: continue
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
