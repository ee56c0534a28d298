# developer builds of the K-outer resident-queries kernel against each other (make dev EXTRA=-DQAMD_RK_...; copies under tools/lib)
for lib in ${LIBS:-libquantization_amd_dev.so rk_W12.so}; do
  for cfg in ${CFGS:-"257:QAMD_NOP=1" "1024:QAMD_RQ_GROUPS=8" "520:QAMD_GEMM_CFG=s"}; do
    Q=${cfg%%:*}; EX=${cfg#*:}
    env QAMD_LIB_PATH="tools/lib/$lib" $EX python3 bench.py --batch-queries $Q --k 30 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$lib', 'Q=$Q', round(r['ms_per_step'],3))" || exit 1
  done
done
