#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
rm -f gpurun_out/r3c5_ab.log
(timeout 900 python -m pytest tests/test_rowfree_gpu.py tests/test_runtime_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r3c5_tests.log 2>&1
for dbg in 0 2 4; do
  echo "== OMNI_GEMV_DBG=$dbg" >> gpurun_out/r3c5_ab.log
  OMNI_GEMV_DBG=$dbg tools/gpu_prof_cmd.sh r3c5_d$dbg python $R/bench.py --steps 32 --warmup 4 --no-extras --fused-level 3 2>&1 | grep -E "gemv_kernel|attn_merge|general_norm_v2|flash" | cut -c1-140 >> gpurun_out/r3c5_ab.log
done
for lvl in 3 2; do
  python bench.py --no-extras --fused-level $lvl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('level', d['config']['fused_ext_level'], d['ms_per_step'], 'ms', d['value'], 'tok/s')" >> gpurun_out/r3c5_ab.log 2>&1
done
