#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c57; O=gpurun_out/r3c57; rm -f $O/*.log
for mb in 0 8 12 16 20 24 0 8 12 16 20; do
  echo "bs64 prefetch_mb=$mb $(OMNI_PREFETCH_MB=$mb timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | sed -n 2p | cut -c1-60)" >> $O/steps.log
done
cat $O/steps.log
