#!/bin/bash
# Developer helper, runs ON THE GPU BOX: per-kernel times of one python tool under rocprofv3.
#   tools/kstats.sh <out tag> <tool.py> [args...]   ->  gpurun_out/<tag>_kstats.txt
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/kstats_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- python3 "$@" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
grep "tree build\|QAMD" "$OUT/run.log"
python3 - "$OUT" "gpurun_out/${TAG}_kstats.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
with open(sys.argv[2], "w") as o:
    for r in list(csv.DictReader(open(f)))[:16]:
        line = f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} min_us {float(r["MinNs"])/1e3:9.1f} pct {r["Percentage"]}'
        print(line); o.write(line + "\n")
PY
