#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c78; O=gpurun_out/r3c78; rm -f $O/*.log
(timeout 600 python -m pytest tests/test_sparse_utils_gpu.py tests/test_lserve_runtime_gpu.py tests/test_reference_lserve_layer_golden_gpu.py -x -q 2>&1 | tail -2) > $O/tests.log 2>&1
for lib in head new head new; do
  if [ $lib = new ]; then unset OMNI_TUNE_LIB; else export OMNI_TUNE_LIB=tune_libs/libhead.so; fi
  echo "$lib $(timeout 300 python tools/kernel_bench.py kv 2>&1 | grep -v amdgpu.ids | grep -i "select\|topk" | cut -c17-120)" >> $O/tests.log
done
unset OMNI_TUNE_LIB
echo "lserve $(timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/tests.log
cat $O/tests.log
