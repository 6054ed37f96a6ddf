#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c72; O=gpurun_out/r3c72; rm -f $O/*.log
for v in 1 0 1 0; do
  echo "arm_o=$v $(OMNI_LSERVE_ARM_O=$v timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/steps.log
