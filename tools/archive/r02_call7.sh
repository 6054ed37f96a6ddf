#!/bin/bash
# round 2, GPU call 7: weight-ring depth of the 64-row GEMV tile (4 / 8 / 16 k-steps in flight per wave)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for R in 4 8 16; do
  echo "== ring $R" >> gpurun_out/c7_sweep.log
  OMNI_TUNE_LIB=tune_libs/lib_ring$R.so timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c7_sweep.log 2>&1
done
cat gpurun_out/c7_sweep.log
