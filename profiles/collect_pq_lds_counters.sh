export TMPDIR=/tmp PART=pq
OUT=gpurun_out/pql
rm -rf $OUT; mkdir -p $OUT
i=0
for CTRS in "SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH" "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 tools/prof_bin_pq.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $CTRS"; tail -2 $OUT/p$i.log; }
done
python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("gpurun_out/pql/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pq_scan" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v)/len(v):.4e}")
PY
