#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_elementwise_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_lserve_runtime_gpu.py -q -k "w8 or W8 or lserve" > gpurun_out/w8_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/w8_tests.log; tail -4 gpurun_out/w8_tests.log
echo "== old"; OMNI_TUNE_LIB=tune_libs/lib_w8old.so timeout 200 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids
echo "== new"; timeout 200 python tools/lserve_steps.py kv8 32 2>&1 | grep -v amdgpu.ids
