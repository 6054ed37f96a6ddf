#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; O=gpurun_out/r3c15; mkdir -p $O
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -5) > $O/tests_gemm.log 2>&1
for mode in chn w8; do
  timeout 300 python tools/gemm_ab.py $mode > $O/ab_dma_$mode.log 2>&1
done
OMNI_GEMM_EXACT=3 timeout 300 python tools/gemm_ab.py grp > $O/ab_dma_grp.log 2>&1
tail -n 7 $O/*.log | cut -c1-150
