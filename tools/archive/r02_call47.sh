#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_elementwise_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_lserve_runtime_gpu.py tests/test_runtime_gpu.py tests/test_fine_grained_gpu.py tests/test_per_tensor_kv8_gpu.py -q > gpurun_out/nosum_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/nosum_tests.log; tail -6 gpurun_out/nosum_tests.log | grep -v "^E   +\|^E            +"
timeout 200 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids
