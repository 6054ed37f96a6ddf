#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c63; O=gpurun_out/r3c63; rm -f $O/*.log
for bl in 240 128 192 248 240 192; do
  echo "blocks=$bl $(OMNI_PREFETCH_BLOCKS=$bl timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-60)" >> $O/steps.log
done
cat $O/steps.log
