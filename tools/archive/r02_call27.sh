#!/bin/bash
# off-path overloads: parity tests + HBM rates
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_offpath_gpu.py -x -q > gpurun_out/offpath_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/offpath_tests.log
tail -30 gpurun_out/offpath_tests.log
timeout 300 python tools/offpath_bench.py > gpurun_out/offpath_bench.log 2>&1; tail -20 gpurun_out/offpath_bench.log
