#!/bin/bash
# Developer helper (GPU box): builds tests/c_abi/sharded_threads.c against the dev library (phase timing of the
# sharded top-k on stderr) and runs it with several lane counts / HW queue limits.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p /tmp/devlib && cp tools/lib/libquantization_amd_dev.so /tmp/devlib/libquantization_amd.so
gcc -std=gnu99 -O1 -Iinclude tests/c_abi/sharded_threads.c -L/tmp/devlib -lquantization_amd -Wl,-rpath,/tmp/devlib -lpthread -o /tmp/st
for hq in 4 8 16; do for lanes in 2 3; do echo "hwq $hq lanes $lanes"; GPU_MAX_HW_QUEUES=$hq QAMD_SHARD_LANES=$lanes /tmp/st 2>&1 | grep -v "^topk_common" | tail -3; done; done
