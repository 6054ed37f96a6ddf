#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_elementwise_gpu.py tests/test_runtime_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_reference_lserve_layer_golden_gpu.py tests/test_lserve_runtime_gpu.py -q > gpurun_out/embed_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/embed_tests.log; tail -4 gpurun_out/embed_tests.log
timeout 300 python bench.py --no-extras 2>&1 | grep -v amdgpu.ids | cut -c1-220
