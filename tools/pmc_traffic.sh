#!/bin/bash
# HBM traffic of the scan kernels at the model and micro shapes, for profiles/traffic.json: separate rocprofv3 --pmc
# passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM: FETCH_SIZE counts half of a wide coalesced read on
# gfx950, so traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024).  "train" = the forward as the training step runs it
# (it also writes the backward's state checkpoints); plain "scan_fwd" = the inference forward (no checkpoints).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_traffic
rm -rf $out; mkdir -p $out
run() { # tag, counter, modes, shape
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $out/$1 -- python tools/bench_scan.py --shapes $4 --iters 4 --modes $3 > $out/$1.log 2>&1
  rc=$?; echo "[$1] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
for shp in 64x768x1024 256x768x128; do
  run fetch_inf_$shp FETCH_SIZE fwd $shp
  run write_inf_$shp WRITE_SIZE fwd $shp
  run fetch_trn_$shp FETCH_SIZE bwd $shp
  run write_trn_$shp WRITE_SIZE bwd $shp
done
python - <<'PY'
import csv, glob, collections, json, hashlib, os, subprocess
def mean(tag, counter, pat):
    v = []
    for f in glob.glob(f"gpurun_out/pmc_traffic/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    v = v[1:] if len(v) > 2 else v
    return sum(v) / len(v) if v else None
res = {}
for shp in ("64x768x1024", "256x768x128"):
    key = shp + "x16"
    for name, tag, pat in (("scan_fwd", "inf", "scan_fwd_seq_kernel"), ("scan_fwd_train", "trn", "scan_fwd_seq_kernel"),
                           ("scan_bwd", "trn", "scan_bwd")):
        f, w = mean(f"fetch_{tag}_{shp}", "FETCH_SIZE", pat), mean(f"write_{tag}_{shp}", "WRITE_SIZE", pat)
        if f is not None and w is not None:
            res[f"{name}:{key}"] = int(2 * f * 1024 + w * 1024)
            print(f"{name}:{key}  FETCH_SIZE {f:.4g} KiB  WRITE_SIZE {w:.4g} KiB  -> {res[f'{name}:{key}'] / 1e6:.1f} MB")
h = hashlib.sha256()
for n in ("common.h", "scan_common.h", "scan_xlane.h", "scan_fwd.hip", "scan_fwd_seq.hip", "scan_bwd.hip", "scan_bwd_seq.hip"):
    h.update(open(os.path.join("si_mamba_amd", "csrc", n), "rb").read())
res["_sources_sha256"] = h.hexdigest()
res["_measured"] = "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE passes of tools/bench_scan.py, fp32; traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024"
json.dump(res, open("gpurun_out/traffic_new.json", "w"), indent=1)
PY
find $out -name "*.csv" -delete
