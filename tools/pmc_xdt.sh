#!/bin/bash
# PMC passes for the x_proj -> dt_proj kernels through tools/xdt_probe (no torch): LIBS="a.so b.so" bash tools/pmc_xdt.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_xdt
rm -rf $OUT; mkdir -p $OUT
LIBS=${LIBS:-si_mamba_amd/libsimamba_hip.so}
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p$i -- ./tools/xdt_probe $LIBS > $OUT/p$i.log 2>&1
  rc=$?; echo "[pmc pass $i] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_xdt/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xdt" not in k: continue
        key = ("conv " if "true" in k or "Lb1" in k else "plain") + " grid=" + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
find $OUT -name "*.csv" -delete
