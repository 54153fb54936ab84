#!/bin/bash
# PMC passes for the two scan backward kernels at the model shape (tools/dev_bwd_seq.py --time --quick runs both).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_bwd
rm -rf $out; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -- python tools/dev_bwd_seq.py --time --quick > $out/p$i.log 2>&1
  rc=$?; echo "[pmc pass $i] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_bwd/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "simamba" not in k: continue
        agg[k.split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_bwd/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "simamba" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-44:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    v = dur.get(k, [0])
    print("==", k, f"dur_us(median, under PMC)={sorted(v)[len(v)//2]:.1f}")
    for c, v in sorted(d.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
find gpurun_out/pmc_bwd -name "*.csv" -delete
