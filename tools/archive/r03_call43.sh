#!/bin/bash
# decode attention: one tile per load batch (148 VGPRs -> three workgroups per CU) vs two (197 VGPRs, two per CU)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c43; O=gpurun_out/r3c43; rm -f $O/*.log
for lib in new fb1 fb1s768 new fb1 fb1s768; do
  if [ $lib = new ]; then unset OMNI_TUNE_LIB; else export OMNI_TUNE_LIB=tune_libs/lib$lib.so; fi
  echo "== $lib" >> $O/kv.log
  timeout 300 python tools/kernel_bench.py kv kv8 2>&1 | grep -v amdgpu.ids | grep -i "decode" | cut -c17-130 >> $O/kv.log
  echo "$lib $(timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ')" >> $O/steps.log
done
(OMNI_TUNE_LIB=tune_libs/libfb1s768.so timeout 900 python -m pytest tests/test_kv4_gpu.py tests/test_fine_grained_gpu.py tests/test_per_tensor_kv8_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -2) > $O/tests.log 2>&1
cat $O/kv.log $O/steps.log $O/tests.log
