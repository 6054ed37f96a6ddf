#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c61; O=gpurun_out/r3c61; rm -f $O/*.log
for mb in 0 32 48 64 24 32 48; do
  echo "lserve prefetch_mb=$mb $(OMNI_LSERVE_PREFETCH_MB=$mb timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/steps.log
