#!/bin/bash
# mid-M GEMV: refill loads fenced in place (new) vs sunk by the scheduler (nofence)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c22; O=gpurun_out/r3c22
(timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_runtime_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep_fence.log
OMNI_SWEEP_OVERRIDES=0 OMNI_TUNE_LIB=tune_libs/libnofence.so timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids > $O/sweep_nofence.log
timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids > $O/tp_fence.log
cat $O/tests.log; paste -d'|' $O/sweep_fence.log $O/sweep_nofence.log | cut -c1-230; cat $O/tp_fence.log
