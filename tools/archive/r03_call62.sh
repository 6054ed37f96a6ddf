#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c62; O=gpurun_out/r3c62; rm -f $O/*.log
for v in 0 1 0 1; do
  echo "lserve pf_down=$v $(OMNI_LSERVE_PF_DOWN=$v timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1) | $(OMNI_LSERVE_PF_DOWN=$v timeout 300 python tools/lserve_steps.py kv4 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
(timeout 600 python -m pytest tests/test_lserve_runtime_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_runtime_gpu.py tests/test_tp_gpu.py -x -q 2>&1 | tail -2) >> $O/steps.log
cat $O/steps.log
