#!/bin/bash
# A/B of the query-streaming kernel's forms on 768-byte rows (QAMD_QS_FORM 0 = one 128-row block, registers -> LDS;
# 2 = two 96-row slabs by LDS-DMA, operand requested two K-blocks ahead): the batched tests with the kernel forced for
# every batch size, then whole-call times by batch size against the default selection.
# usage: tools/run_qs_form_ab.sh [--help]
if [ "$1" = "--help" ] || [ "$1" = "-h" ]; then sed -n 2,5p "$0"; exit 0; fi
set -e
QAMD_GEMM_CFG=q QAMD_QS_FORM=2 timeout -k 10 600 python -m pytest tests/test_gpu_u8_batch.py -x -q 2>&1 | tail -3
SIZES=129,256,320,384,400,448,512,576,640,703,704,768,1024,2048
for form in 0 2; do
  QAMD_GEMM_CFG=q QAMD_QS_FORM=$form timeout -k 10 300 python tools/time_batch.py $SIZES 2>/dev/null | sed "s/^/form $form forced qs | /"
done
QAMD_GEMM_CFG=r timeout -k 10 300 python tools/time_batch.py 129,256,320,384,400,448,512,576,640,703 2>/dev/null | sed "s/^/row-streaming forced | /"
timeout -k 10 300 python tools/time_batch.py $SIZES 2>/dev/null | sed "s/^/default selection | /"
