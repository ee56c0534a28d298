#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): one bench.py line per quantizer / shape of the final build.
#   profiles/collect_lines.sh <round tag>     ->  gpurun_out/<tag>_bench_line_*.json
set -u
TAG=${1:-r02}
run() { NAME=$1; shift; python3 bench.py --no-cpu-baseline "$@" 2> "gpurun_out/${TAG}_bench_line_$NAME.err" | tail -1 > "gpurun_out/${TAG}_bench_line_$NAME.json" \
        || { tail -3 "gpurun_out/${TAG}_bench_line_$NAME.err"; exit 1; }; cut -c1-260 "gpurun_out/${TAG}_bench_line_$NAME.json"; }
run u8l2 --distance l2
run u8_1536 --dim 1536 --rows 12500000
run bin --quantizer binary --dim 1024 --rows 50000000
run pq --quantizer pq
run batch1024 --batch-queries 1024 --k 30 --steps 5 --warmup 3
run batch1024_1536 --batch-queries 1024 --k 30 --steps 5 --warmup 3 --dim 1536 --rows 12500000
run batch64 --batch-queries 64 --k 30 --steps 10 --warmup 5
run bin_batch64 --quantizer binary --dim 1024 --rows 50000000 --batch-queries 64 --k 30 --steps 10 --warmup 5
run bin_batch1024 --quantizer binary --dim 1024 --rows 50000000 --batch-queries 1024 --k 30 --steps 3 --warmup 2
