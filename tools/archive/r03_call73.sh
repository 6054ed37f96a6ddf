#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c73; O=gpurun_out/r3c73; rm -f $O/*.log
for v in 1 0 1 0; do
  echo "arm_o=$v $(OMNI_ARM_O=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ') | tp $(OMNI_ARM_O=$v timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/steps.log
