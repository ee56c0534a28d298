#!/bin/bash
# Runs ON THE GPU BOX: kernel time + HBM read bytes of the PQ scan (tools/prof_bin_pq.py, PART=pq).  Development aid.
export TMPDIR=/tmp PART=pq
OUT=gpurun_out/pqq
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 tools/prof_bin_pq.py > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o p -- python3 tools/prof_bin_pq.py > $OUT/pmc.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pqq/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:70], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3)
f = glob.glob("gpurun_out/pqq/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "pq_scan" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc.setdefault(r["Kernel_Name"][:50], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "HBM read GB per launch", sum(v) / len(v) * 1024 * 2 / 1e9)
PY
