#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): evidence for the many-queries-at-once kernels of csrc/u8_batch.hip.
#   profiles/collect_batch.sh <round tag, e.g. r02>
# Writes under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_batch_kernel_stats.txt     per-kernel times of topk_batch(30) at 16 / 128 / 256 / 1024 queries, 10M x 768
#   <tag>_batch_counters_1024.txt    PMC counters of the 1024-query main kernel (separate --pmc passes, kernel trace only)
#   <tag>_mfma_int8_ceiling.txt      back-to-back MFMA issue rate of this box (dev library)
#   <tag>_row_stream_patterns.txt    pure-load access-pattern sweep behind the row-streaming kernel (dev library)
#   <tag>_qs_timeline.txt            per-phase cycles of the query-streaming kernel (dev library)
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_batch_prof
mkdir -p "$OUT"
: > "gpurun_out/${TAG}_batch_kernel_stats.txt"
for NQ in 16 128 256 1024; do
    rm -rf "$OUT/stats_$NQ"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$NQ" -o k -- python3 tools/time_batch.py $NQ \
        > "$OUT/stats_$NQ.log" 2>&1 || { tail -5 "$OUT/stats_$NQ.log"; exit 1; }
    { echo "== topk_batch(30), $NQ queries, 10M x 768 (tools/time_batch.py $NQ under rocprofv3 --kernel-trace --stats)"; grep "tree build" "$OUT/stats_$NQ.log";
      python3 - "$OUT/stats_$NQ" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    name = r["Name"].replace("(anonymous namespace)::", "")[:96]
    if name.startswith("void at::") or "quantize16" in name or "minmax" in name:
        continue  # store construction, not the call
    print(f'{name:96s} calls {r["Calls"]:>5s}  avg_us {float(r["AverageNs"])/1e3:9.1f}  min_us {float(r["MinNs"])/1e3:9.1f}')
PY
    } >> "gpurun_out/${TAG}_batch_kernel_stats.txt"
done
PASS=0
CSVS=""
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    PASS=$((PASS + 1))
    rm -rf "$OUT/pmc_$PASS"
    NQ=1024 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pmc_$PASS" -o p -- python3 tools/prof_batch.py \
        > "$OUT/pmc_$PASS.log" 2>&1 || { echo "counter pass $PASS ($CTRS) failed:"; tail -3 "$OUT/pmc_$PASS.log"; continue; }
    F=$(find "$OUT/pmc_$PASS" -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && CSVS="$CSVS $F"
done
set -- $CSVS
FIRST=$1; shift
python3 profiles/summarize.py counters "$FIRST" "${QS_KERNEL:-u8_gemm_qs16_kernel<1}" "gpurun_out/${TAG}_batch_counters_1024.txt" "$@"
export QAMD_LIB_PATH=$PWD/tools/lib/libquantization_amd_dev.so
python3 tools/mfma_peak.py 2>&1 | grep -v amdgpu.ids > "gpurun_out/${TAG}_mfma_int8_ceiling.txt"
python3 tools/tune_stream.py 2>&1 | grep -v amdgpu.ids > "gpurun_out/${TAG}_row_stream_patterns.txt"
QAMD_GEMM_CFG=q python3 tools/gemm_timeline.py 1024 10000000 768 2>&1 | grep -v amdgpu.ids > "gpurun_out/${TAG}_qs_timeline.txt"
tail -n +1 "gpurun_out/${TAG}_batch_kernel_stats.txt" "gpurun_out/${TAG}_mfma_int8_ceiling.txt" "gpurun_out/${TAG}_qs_timeline.txt"
