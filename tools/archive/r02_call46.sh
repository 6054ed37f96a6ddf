#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prefill_attn_gpu.py -q -k "xcd_maps" > gpurun_out/xcd_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/xcd_tests.log; tail -8 gpurun_out/xcd_tests.log | grep -v "^E   +\|^E            +"
