#!/bin/bash
# decode attention with the slimmer softmax (exp2 domain, masks only in a split's last tile, f32 row sums, packed conversions): parity, then A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c42; O=gpurun_out/r3c42
(timeout 900 python -m pytest tests/test_kv4_gpu.py tests/test_fine_grained_gpu.py tests/test_per_tensor_kv8_gpu.py tests/test_edge_cases_gpu.py tests/test_rowfree_gpu.py tests/test_runtime_gpu.py tests/test_lserve_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_reference_lserve_layer_golden_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | grep "rel L2\|passed\|failed\|Error" | tail -12) > $O/tests.log 2>&1
for lib in head new head new; do
  if [ $lib = head ]; then export OMNI_TUNE_LIB=tune_libs/libhead.so; else unset OMNI_TUNE_LIB; fi
  echo "== $lib" >> $O/kv.log
  timeout 300 python tools/kernel_bench.py kv kv8 2>&1 | grep -v amdgpu.ids | grep -i "decode" | cut -c1-200 >> $O/kv.log
  echo "$lib $(timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' ')" >> $O/steps.log
done
cat $O/tests.log $O/kv.log $O/steps.log
