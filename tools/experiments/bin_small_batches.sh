# binary topk_batch of 1 .. 16 queries, 50M x 1024: the vector-ALU multi-query scan against the matrix-core path (developer build)
export QAMD_LIB_PATH=tools/lib/libquantization_amd_dev.so
for Q in 2 3 4 5 6 8 10 11 12 16; do
  for MIN in 100 2; do
    QAMD_BIN_MFMA_MIN=$MIN python3 bench.py --quantizer binary --dim 1024 --rows 50000000 --batch-queries $Q --k 30 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('Q=$Q', 'matrix cores' if $MIN == 2 else 'vector ALU  ', round(r['ms_per_step'],3), 'ms')" || exit 1
  done
done
