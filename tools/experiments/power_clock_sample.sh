# What the part's power and shader clock do under different loads (rocm-smi sampled twice a second while each runs): the 1024-query
# u8 batch on both kernels that serve it, the plain u8 scan, the bare int8 MFMA loops (tools/mfma_peak.py, developer build).
watch_job() {  # $1 label, $2 pid
  while kill -0 $2 2>/dev/null; do
    sleep 0.5
    echo "$1 | $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | sed 's/GPU\[0\]//; s/\t//g' | tr -s ' ' | tr '\n' ';')"
  done
}
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "Max Graphics" | head -1
python3 bench.py --batch-queries 1024 --k 30 --steps 1200 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &
watch_job "batch 1024, query-streaming kernel" $!
QAMD_LIB_PATH=tools/lib/libquantization_amd_dev.so QAMD_GEMM_CFG=s QAMD_RQ_GROUPS=8 python3 bench.py --batch-queries 1024 --k 30 --steps 1200 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 &
watch_job "batch 1024, resident-queries kernel" $!
python3 bench.py --steps 6000 --warmup 5 --no-cpu-baseline > /dev/null 2>&1 &
watch_job "u8 scan" $!
QAMD_LIB_PATH=tools/lib/libquantization_amd_dev.so python3 tools/mfma_peak.py > gpurun_out/mfma_peak.log 2>&1 &
watch_job "bare MFMA loops" $!
tail -5 gpurun_out/mfma_peak.log
