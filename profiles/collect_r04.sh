#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): every BASELINE config's bench line with the headline's evidence
# (kernel stats, PMC traffic, CPU baseline).   profiles/collect_r04.sh <tag> <commit> [names...]
set -u
TAG=${1:-r04}; COMMIT=${2:-unknown}; shift 2 || true
want() { [ $# -eq 0 ] && return 0; }
sel="$*"
run() { n=$1; if [ -z "$sel" ] || [[ " $sel " == *" $n "* ]]; then bash profiles/collect_config.sh "$TAG" "$COMMIT" "$@" || exit 1; fi; }
# (line_only: the bench line alone - a PQ batch is the single scan's kernel once per query; its line quotes that scan's PMC profile x queries)
line_only() { n=$1; shift 2; if [ -z "$sel" ] || [[ " $sel " == *" $n "* ]]; then python3 bench.py "$@" 2> gpurun_out/${TAG}_line_$n.err | tail -1 > gpurun_out/${TAG}_bench_line_$n.json || exit 1; fi; }
#   name        key                 needle                         rows      row B  launches  score B -- bench args
run u8          u8_scan             u8_scan_kernel                 10000000  772    1 4 --
run u8l2        u8_scan             u8_scan_kernel                 10000000  772    1 4 -- --distance l2
run u8_1536     u8_scan_1536        u8_scan_kernel                 12500000  1540   1 4 -- --dim 1536 --rows 12500000
run bin         bin_scan            bin_scan_kernel                50000000  128    1 4 -- --quantizer binary --dim 1024 --rows 50000000
run pq          pq_scan_m96         pq_scan_skew_kernel            10000000  96     1 4 -- --quantizer pq
run pq192       pq_scan_m192        pq_scan_skew_kernel            12500000  192    2 4 -- --quantizer pq --dim 1536 --rows 12500000
run pq48        pq_scan_m48         pq_scan_skew_kernel            20000000  48     1 4 -- --quantizer pq --pq-chunk 16 --rows 20000000
run batch1024   u8_batch1024_768    u8_gemm_qs16_kernel           10000000  772    1 0 -- --batch-queries 1024 --k 30 --steps 5 --warmup 3
run batch1024_1536 u8_batch1024_1536 u8_gemm_qs16_kernel          12500000  1540   1 0 -- --batch-queries 1024 --k 30 --steps 5 --warmup 3 --dim 1536 --rows 12500000
run batch64     u8_batch64_768      u8_gemm_rs_kernel             10000000  772    1 0 -- --batch-queries 64 --k 30 --steps 10 --warmup 5
run bin_batch64 bin_batch64_1024    bin_gemm_rs4_kernel           50000000  128    1 0 -- --quantizer binary --dim 1024 --rows 50000000 --batch-queries 64 --k 30 --steps 10 --warmup 5
run bin_batch1024 bin_batch1024_1024 bin_gemm_rs4_kernel          50000000  128    4 0 -- --quantizer binary --dim 1024 --rows 50000000 --batch-queries 1024 --k 30 --steps 3 --warmup 2
run bin_batch128 bin_batch128_1024  bin_gemm_rs4_kernel           50000000  128    1 0 -- --quantizer binary --dim 1024 --rows 50000000 --batch-queries 128 --k 30 --steps 10 --warmup 5
run bin_batch256 bin_batch256_1024  bin_gemm_rs4_kernel           50000000  128    1 0 -- --quantizer binary --dim 1024 --rows 50000000 --batch-queries 256 --k 30 --steps 10 --warmup 5
run bin_batch512 bin_batch512_1024  bin_gemm_rs4_kernel           50000000  128    2 0 -- --quantizer binary --dim 1024 --rows 50000000 --batch-queries 512 --k 30 --steps 5 --warmup 3
run batch192    u8_batch192_768     u8_gemm_rk16_kernel           10000000  772    1 0 -- --batch-queries 192 --k 30 --steps 10 --warmup 5
run batch384    u8_batch384_768     u8_gemm_rk16_kernel           10000000  772    1 0 -- --batch-queries 384 --k 30 --steps 10 --warmup 5
run batch768    u8_batch768_768     u8_gemm_rk16_kernel           10000000  772    1 0 -- --batch-queries 768 --k 30 --steps 8 --warmup 3
run batch256    u8_batch256_768     u8_gemm_qr16_kernel           10000000  772    1 0 -- --batch-queries 256 --k 30 --steps 10 --warmup 5
line_only pq_batch64 -- --quantizer pq --batch-queries 64 --k 30 --steps 3 --warmup 1
line_only pq_batch1024_1536 -- --quantizer pq --dim 1536 --rows 12500000 --batch-queries 1024 --k 30 --steps 2 --warmup 1
