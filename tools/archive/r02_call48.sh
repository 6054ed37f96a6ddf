#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_tp_gpu.py tests/test_ckpt_gpu.py tests/test_persistent_gpu.py -q > gpurun_out/g128_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g128_tests.log; tail -6 gpurun_out/g128_tests.log | grep -v "^E   +\|^E            +"
timeout 300 python tools/step_ab.py 2>&1 | grep -v amdgpu.ids
