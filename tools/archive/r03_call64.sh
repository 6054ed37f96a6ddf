#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c64; O=gpurun_out/r3c64; rm -f $O/*.log
for bl in 64 96 128 160 192 240 96 128 160; do
  echo "blocks=$bl $(OMNI_PREFETCH_BLOCKS=$bl timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-60)" >> $O/steps.log
done
cat $O/steps.log
