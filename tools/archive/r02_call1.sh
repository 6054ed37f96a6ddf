#!/bin/bash
# round 2, GPU call 1: parity suite with the new cases, prefetch A/B, per-kernel table of the decode step
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_runtime_gpu.py 2>&1 | tail -70 > gpurun_out/c1_pytest.log
timeout 300 python -m pytest tests/test_runtime_gpu.py -m gpu -q --tb=short 2>&1 | tail -40 > gpurun_out/c1_pytest_runtime.log
timeout 600 python tools/decode_ab.py --budgets 8,16,24,32 --policies 1,0 > gpurun_out/c1_ab.log 2>&1
timeout 300 python tools/decode_ab.py --budgets 24 --policies 1 --blocks 112,496 >> gpurun_out/c1_ab.log 2>&1
OMNI_PREFETCH_MB=24 tools/gpu_prof_cmd.sh c1_pf python bench.py --steps 32 --warmup 4 --no-extras > gpurun_out/c1_prof_pf.log 2>&1
OMNI_PREFETCH_MB=0 tools/gpu_prof_cmd.sh c1_nopf python bench.py --steps 32 --warmup 4 --no-extras > gpurun_out/c1_prof_nopf.log 2>&1
timeout 300 python tools/tp_rank_steps.py 128 > gpurun_out/c1_tp.log 2>&1
timeout 300 python tools/decode_ab.py --group-size 128 --batch 64 --budgets 24 --policies 1 >> gpurun_out/c1_ab.log 2>&1
cat gpurun_out/c1_pytest.log gpurun_out/c1_pytest_runtime.log gpurun_out/c1_ab.log gpurun_out/c1_tp.log
