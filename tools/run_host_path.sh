#!/bin/bash
# Developer helper (GPU box): plain-C host-in / host-out search latency on small stores (tools/c/host_path.c).
set -e
cd "$GRAFT_REPO_ROOT"
gcc -O2 -std=c99 -Iinclude tools/c/host_path.c -Lquantization_amd -lquantization_amd -Wl,-rpath,$PWD/quantization_amd -o /tmp/host_path
for n in 10000 100000 1000000; do /tmp/host_path $n 768; done
