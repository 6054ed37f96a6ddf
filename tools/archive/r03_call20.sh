#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c20; O=gpurun_out/r3c20
(timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3) > $O/tests_gemm.log 2>&1
for mode in chn w8; do
  timeout 300 python tools/gemm_ab.py $mode 2>&1 | grep -v amdgpu.ids > $O/ab_$mode.log
done
cat $O/*.log | cut -c1-160
