#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates every number DESIGN.md quotes from the CURRENT tree.
#   profiles/collect_all.sh <round tag> <commit>
# Writes gpurun_out/<tag>_*; copy into profiles/ what is to be judged.
TAG=${1:-r03}
COMMIT=${2:-unknown}
echo "== collect.sh (bench under rocprofv3, PMC traffic)"; bash profiles/collect.sh "$TAG" "$COMMIT" > "gpurun_out/${TAG}_collect.log" 2>&1; tail -3 "gpurun_out/${TAG}_collect.log"
echo "== per-kernel bench"; python3 tools/bench_all.py u8 bin pq topk ids encode 2>/dev/null > "gpurun_out/${TAG}_per_kernel_bench.jsonl"; wc -l "gpurun_out/${TAG}_per_kernel_bench.jsonl"
echo "== batch sizes"; BATCH_NQ=16,64,128,129,192,256,257,385,512,768,1024,2048 python3 tools/bench_all.py batch 2>/dev/null > "gpurun_out/${TAG}_bench_topk_batch_by_queries.jsonl"; wc -l "gpurun_out/${TAG}_bench_topk_batch_by_queries.jsonl"
echo "== bench lines"; bash profiles/collect_lines.sh "$TAG" > "gpurun_out/${TAG}_lines.log" 2>&1; tail -2 "gpurun_out/${TAG}_lines.log"
echo "== single-process sharded bench (two logical shards on this one GPU)"
python3 bench.py --single-process --gpus 2 --devices 0,0 2>/dev/null | tail -1 > "gpurun_out/${TAG}_bench_line_single_process_2shards.json"; cut -c1-200 "gpurun_out/${TAG}_bench_line_single_process_2shards.json"
echo "== two ranks on this one GPU over gloo, self-launched"
python3 bench.py --gpus 2 --backend gloo --all-ranks-on-device 0 2>/dev/null | tail -1 > "gpurun_out/${TAG}_bench_line_selflaunch_2ranks_gloo.json"; cut -c1-200 "gpurun_out/${TAG}_bench_line_selflaunch_2ranks_gloo.json"
echo "== small stores"; { bash tools/run_host_path.sh; python3 tools/time_small.py; python3 tools/time_sharded.py; python3 tools/time_point.py; } 2>/dev/null > "gpurun_out/${TAG}_small_store_latency.txt"; cat "gpurun_out/${TAG}_small_store_latency.txt"
echo "== bursts"; python3 tools/time_bursts.py 2>/dev/null > "gpurun_out/${TAG}_bursts.jsonl"; wc -l "gpurun_out/${TAG}_bursts.jsonl"
echo "== sharded handle, six search threads (tests/c_abi/sharded_threads.c)"
gcc -std=gnu99 -O1 -Iinclude tests/c_abi/sharded_threads.c -Lquantization_amd -lquantization_amd -Wl,-rpath,$PWD/quantization_amd -lpthread -o /tmp/st && /tmp/st > "gpurun_out/${TAG}_sharded_threads.txt" 2>&1; cat "gpurun_out/${TAG}_sharded_threads.txt"
