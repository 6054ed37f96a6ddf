#!/bin/bash
# round 3, call 14: exact-shape prefill kernel (LDS-DMA activations) -- parity + A/B vs head / register-staged exact form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; O=gpurun_out/r3c14; mkdir -p $O
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -5) > $O/tests_gemm.log 2>&1
for mode in chn grp w8; do
  timeout 300 python tools/gemm_ab.py $mode > $O/ab_dma_$mode.log 2>&1
  OMNI_GEMM_EXACT=2 timeout 300 python tools/gemm_ab.py $mode > $O/ab_reg_$mode.log 2>&1
  OMNI_GEMM_EXACT=0 timeout 300 python tools/gemm_ab.py $mode > $O/ab_gen_$mode.log 2>&1
done
tail -n 7 $O/*.log | cut -c1-150
