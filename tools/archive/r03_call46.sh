#!/bin/bash
# LServe decode (sparse attention, G = 1 instantiations): one vs two tiles per load batch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c46; O=gpurun_out/r3c46; rm -f $O/*.log
for lib in fb2 new fb2 new; do
  if [ $lib = new ]; then unset OMNI_TUNE_LIB; else export OMNI_TUNE_LIB=tune_libs/lib$lib.so; fi
  echo "$lib $(timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1) | $(timeout 300 python tools/lserve_steps.py kv4 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
  echo "== $lib" >> $O/kv.log
  timeout 300 python tools/kernel_bench.py kv kv8 2>&1 | grep -v amdgpu.ids | grep -i "decode" | grep -i "B=1 \|B=16" | cut -c17-130 >> $O/kv.log
done
cat $O/steps.log $O/kv.log
