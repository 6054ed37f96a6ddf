#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c66; O=gpurun_out/r3c66; rm -f $O/*.log
timeout 600 python tools/step_split_sweep.py 2>&1 | grep -v amdgpu.ids > $O/split.log
for v in "OMNI_DEFERRED_PART=256" "OMNI_DEFERRED_PART=1024" "OMNI_PREFETCH_DELAY=4" "OMNI_DECODE_RT=256" "X=1"; do
  echo "$v $(env $v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-60)" >> $O/knobs.log
done
cat $O/split.log $O/knobs.log
