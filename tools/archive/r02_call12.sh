#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/c12.log
echo "== AR=4 KW=2 (current)" >> gpurun_out/c12.log
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c12.log 2>&1
echo "== AR=2 KW=4" >> gpurun_out/c12.log
OMNI_SWEEP_AR2=1 OMNI_TUNE_LIB=tune_libs/lib_ar2.so timeout 300 python tools/mid_gemv_sweep.py >> gpurun_out/c12.log 2>&1
grep -v amdgpu.ids gpurun_out/c12.log
