#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c41; O=gpurun_out/r3c41; rm -f $O/ab.log
for v in 1 0 1 0; do
  echo "pf_down=$v $(OMNI_L3_PF_DOWN=$v timeout 300 python bench.py --no-extras 2>&1 | grep -v amdgpu.ids | grep metric | cut -c1-200)" >> $O/ab.log
  echo "   step_ab pf_down=$v $(OMNI_L3_PF_DOWN=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1)" >> $O/ab.log
done
cat $O/ab.log
