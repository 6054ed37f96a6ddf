#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/c23.log
echo "== base" >> gpurun_out/c23.log
OMNI_TUNE_LIB=tune_libs/lib_gemm_base.so timeout 300 python tools/gemm4096.py >> gpurun_out/c23.log 2>&1
echo "== persistent" >> gpurun_out/c23.log
timeout 300 python tools/gemm4096.py >> gpurun_out/c23.log 2>&1
timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -6 >> gpurun_out/c23.log
grep -v amdgpu.ids gpurun_out/c23.log
