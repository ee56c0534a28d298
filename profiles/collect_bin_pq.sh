#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 counters for the binary and PQ kernels.
#   profiles/collect_bin_pq.sh <round tag, e.g. r03> <commit>
# One counter set per pass, --kernel-trace only beside --pmc; the program itself comes right after `--`.
# Writes gpurun_out/<tag>_pmc_bin_pq.txt (copy into profiles/ what is to be judged).
set -u
TAG=${1:-r03}
COMMIT=${2:-unknown}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_binpq_prof
mkdir -p "$OUT"
rm -rf "$OUT/stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o k -- python3 tools/prof_bin_pq.py > "$OUT/stats.log" 2>&1 \
    || { tail -5 "$OUT/stats.log"; exit 1; }
PASS=0
CSVS=""
for CTRS in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    PASS=$((PASS + 1))
    rm -rf "$OUT/pmc_$PASS"
    rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pmc_$PASS" -o p -- python3 tools/prof_bin_pq.py \
        > "$OUT/pmc_$PASS.log" 2>&1 || { echo "counter pass $PASS ($CTRS) failed:"; tail -3 "$OUT/pmc_$PASS.log"; continue; }
    F=$(find "$OUT/pmc_$PASS" -name '*counter_collection.csv' | head -1)
    [ -n "$F" ] && CSVS="$CSVS $F"
done
set -- $CSVS
FIRST=$1; shift
RES="gpurun_out/${TAG}_pmc_bin_pq.txt"
{
  echo "rocprofv3 --pmc passes (one counter set per run, --kernel-trace only) over tools/prof_bin_pq.py -- $TAG, commit $COMMIT"
  echo "means per launch, chip totals; FETCH_SIZE / WRITE_SIZE in KiB (gfx950: read bytes = FETCH_SIZE * 1024 * 2)"
  echo
  echo "== per-kernel times of the same driver (rocprofv3 --kernel-trace --stats)"
  python3 - "$OUT/stats" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    name = r["Name"].replace("(anonymous namespace)::", "")[:90]
    if name.startswith("void at::"):
        continue
    print(f'{name:90s} calls {r["Calls"]:>4s}  avg_us {float(r["AverageNs"])/1e3:9.1f}  min_us {float(r["MinNs"])/1e3:9.1f}')
PY
} > "$RES"
for K in "bin_scan_kernel" "bin_scan_multi_kernel" "bin_gemm_rs_kernel" "pq_scan_fast_kernel" "pq_scan_skew_kernel"; do
  echo >> "$RES"; echo "== $K" >> "$RES"
  python3 profiles/summarize.py counters "$FIRST" "$K" "$OUT/tmp_$K.txt" "$@" > /dev/null
  cat "$OUT/tmp_$K.txt" >> "$RES"
done
cat "$RES"
