#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c17
OMNI_TUNE_LIB=tune_libs/libclk.so timeout 300 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c17/timeline.log
cat gpurun_out/r3c17/timeline.log
