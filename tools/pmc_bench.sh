#!/bin/bash
# MFMA utilisation of the kernels of the bench step: PMC passes over `bench.py` (program directly
# after `--`, no tracing besides --kernel-trace), aggregated per kernel:
#   MFMA flops (SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} * 512), MFMA-busy share = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES ...)
# usage: DTYPE=f32|bf16 bash tools/pmc_bench.sh <outdir>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/pmc_bench}
DT=${DTYPE:-f32}
mkdir -p $OUT
i=0
for ctrs in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-headline --no-bf16-step > $OUT/p$i.log 2>&1
  rc=$?; echo "[pmc pass $i] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 - "$OUT" "$DT" <<'PY'
import csv, glob, collections, sys
out, dt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:100]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key); calls[k] += 1
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for k, d in agg.items():
    mops = d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) + d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0)
    if mops == 0:
        continue
    flops = mops * 512
    t = dur[k] * 1e-9
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    gui = d.get("GRBM_GUI_ACTIVE", 0)
    rows.append((flops, k, calls[k], t, busy, gui, d.get("SQ_BUSY_CYCLES", 0)))
rows.sort(reverse=True)
peak = 157.3e12 if dt == "f32" else 2.5e15
with open(out + "/summary.txt", "w") as fo:
    fo.write(f"# MFMA counters of the bench step ({dt}); peak used for the fraction: {peak/1e12:.0f} TF/s dense\n")
    fo.write("# flops = SQ_INSTS_VALU_MFMA_MOPS_* x 512 ; TF/s = flops / summed dispatch duration (profiled run) ;\n")
    fo.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs) -- the MfmaUtil expression of rocprofv3\n")
    for flops, k, n, t, busy, gui, sqb in rows:
        tf = flops / t / 1e12 if t else 0
        util = busy / (gui / 8 * 1024) if gui else 0
        fo.write(f"{k:100s} calls={n:4d} GFLOP/call={flops/n/1e9:8.2f} us/call={t/n*1e6:8.1f} TF/s={tf:7.1f} of_peak={tf*1e12/peak:5.2f} mfma_busy={util:5.2f}\n")
print(open(out + "/summary.txt").read())
PY
