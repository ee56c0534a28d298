#!/bin/bash
# Runs ON THE GPU BOX (through gpurun) and regenerates the rocprofv3 evidence for bench.py's roofline:
#   profiles/collect.sh <round tag, e.g. r02> <commit>
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> <tag>_bench_u8_kernel_stats.csv
# 2. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (counters are never combined with tracing
#    domains other than the kernel trace)                          -> <tag>_pmc_u8_scan.json
# The bench program itself comes right after `--` (python3 bench.py): no env/bash hop under rocprofv3.
set -u
TAG=${1:-r02}
COMMIT=${2:-unknown}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_prof
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --no-cpu-baseline \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || { tail -5 "$OUT/bench_under_rocprof.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 bench.py --no-cpu-baseline \
    --steps 6 --warmup 2 > /dev/null 2> "$OUT/pmc_fetch.err" || { tail -5 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- python3 bench.py --no-cpu-baseline \
    --steps 6 --warmup 2 > /dev/null 2> "$OUT/pmc_write.err" || { tail -5 "$OUT/pmc_write.err"; exit 1; }
STATS=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
FETCH=$(find "$OUT/pmc_fetch" -name '*counter_collection.csv' | head -1)
WRITE=$(find "$OUT/pmc_write" -name '*counter_collection.csv' | head -1)
python3 profiles/summarize.py stats "$STATS" "gpurun_out/${TAG}_bench_u8_kernel_stats.csv"
python3 profiles/summarize.py pmc "$FETCH" "$WRITE" u8_scan_kernel 10000000 772 "gpurun_out/${TAG}_pmc_u8_scan.json" "$COMMIT"
cp "$OUT/bench_under_rocprof.json" "gpurun_out/${TAG}_bench_under_rocprof.json"
