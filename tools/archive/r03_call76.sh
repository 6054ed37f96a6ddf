#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c76; O=gpurun_out/r3c76; rm -f $O/*.log
for v in 1 2 1 2; do
  echo "narrow=$v $(OMNI_GEMV_NARROW=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | sed -n 2p | cut -c1-60)" >> $O/steps.log
done
cat $O/steps.log
