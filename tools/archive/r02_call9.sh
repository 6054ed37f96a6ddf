#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/c9.log
echo "== pinned A loads" >> gpurun_out/c9.log
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c9.log 2>&1
echo "== not pinned" >> gpurun_out/c9.log
OMNI_SWEEP_OVERRIDES=0 OMNI_TUNE_LIB=tune_libs/lib_nopin.so timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c9.log 2>&1
grep -v amdgpu.ids gpurun_out/c9.log
