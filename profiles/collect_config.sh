#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the full evidence for ONE bench.py configuration.
#   profiles/collect_config.sh <tag> <commit> <name> <pmc key> <kernel needle[|more]> <rows per launch> \
#                              <row read bytes> <launches per scan/step> <score bytes written per row> -- <bench.py args>
# 1. rocprofv3 --kernel-trace --stats of the bench command          -> gpurun_out/<tag>_kstats_<name>.csv
# 2. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (never combined with tracing domains other than the
#    kernel trace), summarised with the guide's gfx950 correction     -> profiles/<tag>_pmc_<key>.json (+ copy in gpurun_out)
# 3. the bench line itself, un-profiled, with the CPU baseline; it quotes (2) as roofline.traffic because the kernel
#    source hash matches                                              -> gpurun_out/<tag>_bench_line_<name>.json
# The bench program comes right after `--` (python3 bench.py): no env/bash hop under rocprofv3.
set -u
TAG=$1; COMMIT=$2; NAME=$3; KEY=$4; NEEDLE=$5; ROWS=$6; ROWB=$7; LPU=$8; WB=$9
shift 10
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_prof_${NAME}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --no-cpu-baseline "$@" \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || { tail -5 "$OUT/stats.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 bench.py --no-cpu-baseline "$@" \
    --steps 4 --warmup 2 > /dev/null 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- python3 bench.py --no-cpu-baseline "$@" \
    --steps 4 --warmup 2 > /dev/null 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
STATS=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
FETCH=$(find "$OUT/pmc_fetch" -name '*counter_collection.csv' | head -1)
WRITE=$(find "$OUT/pmc_write" -name '*counter_collection.csv' | head -1)
python3 profiles/summarize.py stats "$STATS" "gpurun_out/${TAG}_kstats_${NAME}.csv"
python3 profiles/summarize.py pmc "$FETCH" "$WRITE" "$NEEDLE" "$ROWS" "$ROWB" "profiles/${TAG}_pmc_${KEY}.json" "$COMMIT" "$KEY" "$LPU" "$WB" > /dev/null \
    || { echo "pmc summary failed for $NAME"; exit 1; }
cp "profiles/${TAG}_pmc_${KEY}.json" "gpurun_out/${TAG}_pmc_${KEY}.json"
tail -1 "$OUT/bench_under_rocprof.json" > "gpurun_out/${TAG}_bench_under_rocprof_${NAME}.json"
python3 bench.py "$@" 2> "$OUT/line.err" | tail -1 > "gpurun_out/${TAG}_bench_line_${NAME}.json" || { tail -5 "$OUT/line.err"; exit 1; }
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write"
python3 - "$NAME" "gpurun_out/${TAG}_bench_line_${NAME}.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
r, c = j["roofline"], j.get("cpu_baseline") or {}
print(f"{sys.argv[1]:18s} value {j['value']:.4g} {j['unit']}  ms/step {j['ms_per_step']:.4f}  frac {r['frac']:.3f} ({r['bound']})  "
      f"traffic {r.get('traffic')}  cpu {c.get('value')} ({c.get('kind')}, parity {c.get('gpu_matches_cpu_bits')})")
PY
