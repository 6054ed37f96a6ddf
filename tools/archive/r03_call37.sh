#!/bin/bash
# level 3: which weights should the norm in front of gate_up prefetch?  gate_up's head (committed) | all of down's (29.6 MB)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c37; O=gpurun_out/r3c37; rm -f $O/ab.log
for rep in 1 2; do
for v in 0 1 2; do
  echo "pf_down=$v $(OMNI_L3_PF_DOWN=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1)" >> $O/ab.log
done
done
cat $O/ab.log
