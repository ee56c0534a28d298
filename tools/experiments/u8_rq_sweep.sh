# u8 topk_batch by batch size, 10M x 768: the resident-queries kernel (u8_gemm_rq16_kernel) against the earlier selection (developer build)
#   QS="129 256 ..."  batch sizes     VARIANTS="0 1 1:QAMD_RQ_STEP=0 ..."  QAMD_RQ value[:extra env assignment]
export QAMD_LIB_PATH=tools/lib/libquantization_amd_dev.so
for Q in ${QS:-129 192 193 256 257 320 384 512 768 1024 1536}; do
  for V in ${VARIANTS:-0 1}; do
    RQ=${V%%:*}; EX=${V#*:}; [ "$EX" = "$V" ] && EX="QAMD_NOP=1"
    env QAMD_RQ=$RQ $EX python3 bench.py --batch-queries $Q --k 30 --steps 8 --warmup 2 --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('Q=$Q', '$V'.ljust(28), round(r['ms_per_step'],3), 'ms  mfma', round(r['roofline']['mfma_frac'],3))" || exit 1
  done
done
