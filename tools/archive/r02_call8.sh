#!/bin/bash
# round 2, GPU call 8: ablations of the 64-row GEMV tile (what bounds M = 64 / 128)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/c8_abl.log
echo "== baseline" >> gpurun_out/c8_abl.log
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c8_abl.log 2>&1
for A in 1 2 4 8 9 13; do
  echo "== ablate $A" >> gpurun_out/c8_abl.log
  OMNI_SWEEP_OVERRIDES=0 OMNI_TUNE_LIB=tune_libs/lib_abl$A.so timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c8_abl.log 2>&1
done
grep -v amdgpu.ids gpurun_out/c8_abl.log
