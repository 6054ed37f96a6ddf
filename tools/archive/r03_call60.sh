#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c60; O=gpurun_out/r3c60; rm -f $O/*.log
for mb in 0 8 16 32 0 16; do
  echo "lserve prefetch_mb=$mb $(OMNI_LSERVE_PREFETCH_MB=$mb timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
echo "tp default $(timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
cat $O/steps.log
