#!/bin/bash
# round 2, GPU call 5: PMC traffic of the four decode GEMVs (separate FETCH / WRITE passes), bench line, LServe fused merge
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_lserve_runtime_gpu.py tests/test_fine_grained_gpu.py tests/test_per_tensor_kv8_gpu.py tests/test_runtime_gpu.py tests/test_ckpt_gpu.py -m gpu -q --tb=short 2>&1 | tail -12 > gpurun_out/c5_pytest.log
timeout 300 python tools/lserve_steps.py > gpurun_out/c5_lserve.log 2>&1
for shape in "28672 4096 16 0" "6144 4096 16 0" "4096 4096 16 1" "4096 14336 16 1"; do
  echo "== gemv N K M deferred = $shape" >> gpurun_out/c5_pmc.log
  tools/gpu_pmc_traffic.sh w4a8_gemv python $R/tools/gemv_loop.py $shape >> gpurun_out/c5_pmc.log 2>&1
done
( time python bench.py ) > gpurun_out/c5_bench.log 2>&1
cat gpurun_out/c5_pytest.log gpurun_out/c5_lserve.log gpurun_out/c5_pmc.log; tail -4 gpurun_out/c5_bench.log | cut -c1-2500
