#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c38; O=gpurun_out/r3c38; rm -f $O/ab.log
for cfg in "0 40" "1 40" "1 30" "1 24" "1 32" "1 48" "0 40" "1 40"; do
  set -- $cfg
  echo "pf_down=$1 mb=$2 $(OMNI_L3_PF_DOWN=$1 OMNI_PREFETCH_MB=$2 timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1)" >> $O/ab.log
done
cat $O/ab.log
