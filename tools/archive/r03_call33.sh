#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c33; O=gpurun_out/r3c33
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids | head -4 > $O/sweep_single.log
OMNI_GEMV_NARROW_PAIR=1 OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids | head -4 > $O/sweep_pair.log
(OMNI_GEMV_NARROW_PAIR=1 timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "70b or mid_batches" 2>&1 | tail -3) > $O/tests_pair.log 2>&1
timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids > $O/tp_single.log
OMNI_GEMV_NARROW_PAIR=1 timeout 300 python tools/tp_rank_steps.py 128 2>&1 | grep -v amdgpu.ids > $O/tp_pair.log
paste -d'|' $O/sweep_single.log $O/sweep_pair.log | cut -c1-200; cat $O/tests_pair.log $O/tp_single.log $O/tp_pair.log
