set -e
timeout -k 10 900 python -m pytest tests/test_gpu_batch_pq_bin.py -x -q 2>&1 | tail -2
QAMD_BIN4_MIN=12 timeout -k 10 900 python -m pytest tests/test_gpu_batch_pq_bin.py -x -q 2>&1 | tail -2
QAMD_BIN4_MIN=12 timeout -k 10 900 python tools/fuzz_bin_batch.py 800 106 2>&1 | grep -v " ok$" | tail -3
timeout -k 10 900 python tools/fuzz_bin_batch.py 400 107 2>&1 | grep -v " ok$" | tail -3
for v in 0 1; do
  for nq in 256 1024; do
  QAMD_BIN4=$v timeout -k 10 300 python bench.py --no-cpu-baseline --quantizer binary --dim 768 --rows 60000000 --batch-queries $nq --k 30 --steps 3 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('dim 768 BIN4=$v ${nq}q', round(d['ms_per_step'],3), 'ms')"
  done
done
