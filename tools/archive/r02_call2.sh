#!/bin/bash
# round 2, GPU call 2: the failing cases of call 1, prefetch A/B, per-kernel tables with / without prefetch
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fine_grained_gpu.py tests/test_prefill_attn_gpu.py tests/test_per_tensor_kv8_gpu.py -m gpu -q --tb=short 2>&1 | tail -30 > gpurun_out/c2_pytest.log
timeout 600 python tools/decode_ab.py --budgets 8,16,24,32 --policies 1,0 > gpurun_out/c2_ab.log 2>&1
timeout 300 python tools/decode_ab.py --budgets 24 --policies 1 --blocks 112,496 >> gpurun_out/c2_ab.log 2>&1
OMNI_PREFETCH_MB=24 tools/gpu_prof_cmd.sh c2_pf python $R/bench.py --steps 32 --warmup 4 --no-extras > gpurun_out/c2_prof_pf.log 2>&1
OMNI_PREFETCH_MB=0 tools/gpu_prof_cmd.sh c2_nopf python $R/bench.py --steps 32 --warmup 4 --no-extras > gpurun_out/c2_prof_nopf.log 2>&1
timeout 300 python tools/decode_ab.py --group-size 128 --batch 64 --budgets 24 --policies 1 --steps 24 >> gpurun_out/c2_ab.log 2>&1
tools/gpu_prof_cmd.sh c2_tp python $R/tools/tp_rank_steps.py 128 > gpurun_out/c2_prof_tp.log 2>&1
cat gpurun_out/c2_pytest.log gpurun_out/c2_ab.log
