#!/bin/bash
# round 3, call 2: parity of the row-kernel-free path + A/B of fusion levels 2 / 3 + per-kernel table of level 3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_rowfree_gpu.py tests/test_runtime_gpu.py tests/test_reference_layer_golden_gpu.py -x -q 2>&1 | tail -25) > gpurun_out/r3c2_tests.log 2>&1
for lvl in 3 2 3 2; do
  python bench.py --no-extras --fused-level $lvl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('level', d['config']['fused_ext_level'], d['ms_per_step'], 'ms', d['value'], 'tok/s')" >> gpurun_out/r3c2_ab.log 2>&1
done
OMNI_DOWN_NT=0 python bench.py --no-extras --fused-level 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('level 3 down plain-loads', d['ms_per_step'], 'ms', d['value'], 'tok/s')" >> gpurun_out/r3c2_ab.log 2>&1
tools/gpu_prof_cmd.sh r3c2_l3 python bench.py --steps 32 --warmup 4 --no-extras --fused-level 3 > gpurun_out/r3c2_prof.log 2>&1
