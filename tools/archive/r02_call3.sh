#!/bin/bash
# round 2, GPU call 3: kernarg-preload probe, prefetch sweeps (blocks / budget / delay), M = 64 / 128 paths, full parity suite
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
{ echo "== default build"; ./tune_libs/kernarg_probe_plain; echo "== -mllvm -amdgpu-kernarg-preload-count=16"; ./tune_libs/kernarg_probe_preload; } > gpurun_out/c3_kernarg.log 2>&1
timeout 900 python tools/decode_ab.py --budgets 24,28,32,40 --policies 0 --blocks 240,496,752 > gpurun_out/c3_ab.log 2>&1
for d in 2 6; do echo "== OMNI_PREFETCH_DELAY=$d" >> gpurun_out/c3_ab.log; OMNI_PREFETCH_DELAY=$d timeout 300 python tools/decode_ab.py --budgets 28 --policies 0 --blocks 496 >> gpurun_out/c3_ab.log 2>&1; done
echo "== llama2_70b tp8 rank bs128, prefetch 0 / 24" > gpurun_out/c3_tp.log
OMNI_PREFETCH_MB=0 timeout 300 python tools/tp_rank_steps.py 128 >> gpurun_out/c3_tp.log 2>&1
OMNI_PREFETCH_MB=24 timeout 300 python tools/tp_rank_steps.py 128 >> gpurun_out/c3_tp.log 2>&1
OMNI_PREFETCH_MB=0 tools/gpu_prof_cmd.sh c3_tp python $R/tools/tp_rank_steps.py 128 > gpurun_out/c3_prof_tp.log 2>&1
OMNI_PREFETCH_MB=0 tools/gpu_prof_cmd.sh c3_g128 python $R/bench.py --group-size 128 --batch 64 --steps 16 --warmup 4 --no-extras > gpurun_out/c3_prof_g128.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -25 > gpurun_out/c3_pytest.log
cat gpurun_out/c3_kernarg.log gpurun_out/c3_ab.log gpurun_out/c3_tp.log gpurun_out/c3_pytest.log
