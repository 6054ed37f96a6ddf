#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_rowfree_gpu.py tests/test_runtime_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r3c4_tests.log 2>&1
for dbg in 0 1 2 4 7; do
  echo "== OMNI_GEMV_DBG=$dbg" >> gpurun_out/r3c4_ab.log
  OMNI_GEMV_DBG=$dbg tools/gpu_prof_cmd.sh r3c4_d$dbg python $R/bench.py --steps 32 --warmup 4 --no-extras --fused-level 3 2>&1 | grep -E "gemv_kernel|attn_merge|general_norm_v2|flash" | cut -c1-140 >> gpurun_out/r3c4_ab.log
done
