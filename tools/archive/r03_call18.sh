#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c18; O=gpurun_out/r3c18
for pr in 0 7 8 9 10; do
  echo "== OMNI_GEMM_PRIO=$pr" >> $O/ab.log
  OMNI_GEMM_PRIO=$pr timeout 300 python tools/gemm_ab.py chn 2>&1 | grep -v amdgpu.ids | head -4 >> $O/ab.log
done
for pr in 0 8; do
  echo "== timeline OMNI_GEMM_PRIO=$pr" >> $O/timeline.log
  OMNI_GEMM_PRIO=$pr OMNI_TUNE_LIB=tune_libs/libclk.so timeout 300 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | head -12 >> $O/timeline.log
done
cat $O/ab.log $O/timeline.log | cut -c1-220
