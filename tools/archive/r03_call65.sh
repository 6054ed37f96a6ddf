#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c65; O=gpurun_out/r3c65; rm -f $O/*.log
for cfg in "160 32" "160 40" "160 48" "240 40" "160 40"; do set -- $cfg
  echo "bs16 blocks=$1 mb=$2 $(OMNI_PREFETCH_BLOCKS=$1 OMNI_PREFETCH_MB=$2 timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-60)" >> $O/steps.log
done
for bl in 240 160 128 240 160; do
  echo "bs64 blocks=$bl $(OMNI_PREFETCH_BLOCKS=$bl timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | sed -n 2p | cut -c1-60) | tp $(OMNI_PREFETCH_BLOCKS=$bl timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1) | lserve $(OMNI_PREFETCH_BLOCKS=$bl timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/steps.log
