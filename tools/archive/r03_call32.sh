#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c32; O=gpurun_out/r3c32
for sp in 0 64 0 64 240 16; do
  echo "== OMNI_SIDE_PREFETCH=$sp" >> $O/step_ab.log
  OMNI_SIDE_PREFETCH=$sp timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1 >> $O/step_ab.log
done
cat $O/step_ab.log
