#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c19; O=gpurun_out/r3c19
# static: slot 1 high at prio 1 / 3; sliced at prio 3 with 2.56 us and 0.64 us slices
for pr in $((1024+32+1)) $((1024+96+1)) $((96+8)) $((96+6)) $((96+11)); do
  echo "== timeline OMNI_GEMM_PRIO=$pr" >> $O/timeline.log
  OMNI_GEMM_PRIO=$pr OMNI_TUNE_LIB=tune_libs/libclk.so timeout 300 python tools/gemm_timeline.py 2>&1 | grep -v amdgpu.ids | head -6 >> $O/timeline.log
done
cat $O/timeline.log | cut -c1-220
