# developer experiment (GPU box): kernel time of bin_gemm_qs4_kernel under timing ablations (results are wrong by construction)
export TMPDIR=/tmp
for V in base 1 2 3; do for Q in 129 256; do
  if [ $V = base ]; then unset QAMD_LIB_PATH; else export QAMD_LIB_PATH=$PWD/tools/lib/libqamd_abl$V.so; fi
  rm -rf /tmp/abl_prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -o p -- python3 bench.py --no-cpu-baseline --quantizer binary --dim 1024 --rows 50000000 --batch-queries $Q --k 30 --steps 3 --warmup 1 > /dev/null 2>&1
  F=$(find /tmp/abl_prof -name '*kernel_stats.csv' | head -1)
  echo "variant $V Q $Q: $(grep 'bin_gemm_qs4_kernel<1' $F | head -1 | awk -F, '{print "calls",$(NF-6),"avg_ns",$(NF-4)}')"
done; done
