#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c23; O=gpurun_out/r3c23
(timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_runtime_gpu.py tests/test_reference_layer_golden_gpu.py -x -q 2>&1 | tail -3) > $O/tests.log 2>&1
timeout 300 python tools/gemm_ab.py grp 2>&1 | grep -v amdgpu.ids > $O/ab_grp_exact.log
OMNI_GEMM_EXACT=0 timeout 300 python tools/gemm_ab.py grp 2>&1 | grep -v amdgpu.ids > $O/ab_grp_generic.log
OMNI_TUNE_LIB=tune_libs/libnofence.so timeout 300 python tools/gemm_ab.py grp 2>&1 | grep -v amdgpu.ids > $O/ab_grp_before.log
OMNI_SWEEP_OVERRIDES=0 timeout 300 python tools/mid_gemv_sweep.py 2>&1 | grep -v amdgpu.ids | tail -4 > $O/sweep.log
cat $O/tests.log; for f in exact generic before; do echo "== $f"; cut -c1-130 $O/ab_grp_$f.log; done; cat $O/sweep.log
