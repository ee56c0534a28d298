export QAMD_LIB_PATH=tools/lib/libquantization_amd_dev.so
for Q in 289 320 384 448 512 576 640 768 864 1024; do
  for P in 1 2 3 4; do
    QAMD_BIN_RS4_PASSES=$P python3 bench.py --quantizer binary --dim 1024 --rows 50000000 --batch-queries $Q --k 30 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('Q=$Q passes<=$P', round(r['ms_per_step'],3), 'ms')" || exit 1
  done
done
