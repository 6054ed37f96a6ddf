#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c29; O=gpurun_out/r3c29
for part in 512 256 512 256; do
  echo "== OMNI_DEFERRED_PART=$part" >> $O/step_ab.log
  OMNI_DEFERRED_PART=$part timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids >> $O/step_ab.log
done
cat $O/step_ab.log
