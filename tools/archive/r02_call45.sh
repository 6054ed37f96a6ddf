#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -k "w8 or W8" > gpurun_out/w8_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/w8_tests.log; tail -3 gpurun_out/w8_tests.log
timeout 200 python tools/gemm_w8.py 2>&1 | grep -v amdgpu.ids
