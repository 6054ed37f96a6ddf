#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c52; O=gpurun_out/r3c52; rm -f $O/*.log
(OMNI_QKV_SLABS=1 timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_lserve_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_persistent_gpu.py tests/test_tp_gpu.py tests/test_ckpt_gpu.py tests/test_kv4_gpu.py tests/test_fine_grained_gpu.py tests/test_per_tensor_kv8_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
echo "tp $(timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1) | off $(OMNI_QKV_SLABS=0 timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/tests.log
cat $O/tests.log
