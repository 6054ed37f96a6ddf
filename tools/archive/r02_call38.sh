#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_layer_golden_gpu.py -q > gpurun_out/golden_g128.log 2>&1; echo "pytest rc=$?" >> gpurun_out/golden_g128.log; tail -25 gpurun_out/golden_g128.log | grep -v "^E   +\|^E            +"
