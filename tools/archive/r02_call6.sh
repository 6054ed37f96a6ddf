#!/bin/bash
# round 2, GPU call 6: MZ=2 in-workgroup M=65..128 GEMV tile, W8A8 deferred split-K epilogue (LServe), bench line
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_elementwise_gpu.py tests/test_lserve_runtime_gpu.py tests/test_runtime_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -15 > gpurun_out/c6_pytest.log
timeout 300 python tools/tp_rank_steps.py 128 > gpurun_out/c6_tp128.log 2>&1
timeout 300 python tools/lserve_steps.py > gpurun_out/c6_lserve.log 2>&1
OMNI_LSERVE_DEFER=0 timeout 300 python tools/lserve_steps.py > gpurun_out/c6_lserve_nodefer.log 2>&1
( time python bench.py ) > gpurun_out/c6_bench.log 2>&1
cat gpurun_out/c6_pytest.log gpurun_out/c6_tp128.log gpurun_out/c6_lserve.log gpurun_out/c6_lserve_nodefer.log; tail -4 gpurun_out/c6_bench.log | cut -c1-2500
