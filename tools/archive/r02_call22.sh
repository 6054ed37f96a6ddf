#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/c22.log
echo "== shipped" >> gpurun_out/c22.log
timeout 300 python tools/gemm4096.py >> gpurun_out/c22.log 2>&1
echo "== setprio" >> gpurun_out/c22.log
OMNI_TUNE_LIB=tune_libs/lib_gemm_prio.so timeout 300 python tools/gemm4096.py >> gpurun_out/c22.log 2>&1
grep -v amdgpu.ids gpurun_out/c22.log
