#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_lserve_layer_golden_gpu.py tests/test_lserve_runtime_gpu.py -q > gpurun_out/lserve_golden.log 2>&1; echo "pytest rc=$?" >> gpurun_out/lserve_golden.log
tail -60 gpurun_out/lserve_golden.log
