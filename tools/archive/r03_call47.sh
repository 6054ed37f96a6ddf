#!/bin/bash
# step-begin kernel, last layer at level 3 (final norm consumes the slabs), slab batches beyond 8: parity, then step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c47; O=gpurun_out/r3c47; rm -f $O/*.log
(timeout 900 python -m pytest tests/test_elementwise_gpu.py tests/test_runtime_gpu.py tests/test_lserve_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_persistent_gpu.py tests/test_tp_gpu.py -x -q 2>&1 | tail -4) > $O/tests.log 2>&1
for v in 1 0 1 0; do
  echo "last_l3=$v $(OMNI_L3_LAST=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ')" >> $O/steps.log
done
echo "lserve $(timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
cat $O/tests.log $O/steps.log
