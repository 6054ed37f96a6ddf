#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c59; O=gpurun_out/r3c59; rm -f $O/*.log
for mb in 16 24 32 40 48 24 32 40; do
  echo "tp rank bs128 prefetch_mb=$mb $(OMNI_PREFETCH_MB=$mb timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/steps.log
