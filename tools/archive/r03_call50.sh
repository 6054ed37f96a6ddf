#!/bin/bash
# qkv projection as slabs consumed by the decode attention: parity through the runners, then A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c50; O=gpurun_out/r3c50; rm -f $O/*.log
(OMNI_QKV_SLABS=1 timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_lserve_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_persistent_gpu.py tests/test_tp_gpu.py tests/test_ckpt_gpu.py -x -q 2>&1 | tail -4) > $O/tests.log 2>&1
for v in 0 1 0 1; do
  echo "qkv_slabs=$v $(OMNI_QKV_SLABS=$v timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-60 | tr '\n' ' ') | tp $(OMNI_QKV_SLABS=$v timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1) | $(OMNI_QKV_SLABS=$v timeout 300 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/steps.log
done
cat $O/tests.log $O/steps.log
