#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_reference_lserve_layer_golden_gpu.py tests/test_lserve_runtime_gpu.py -q > gpurun_out/lserve_golden.log 2>&1; echo "pytest rc=$?" >> gpurun_out/lserve_golden.log; tail -3 gpurun_out/lserve_golden.log
timeout 600 python tools/lserve_prefill.py kv8 16384 65536 256000 > gpurun_out/lserve_prefill.log 2>&1; echo "rc=$?" >> gpurun_out/lserve_prefill.log
tail -5 gpurun_out/lserve_prefill.log
