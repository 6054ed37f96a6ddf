#!/bin/bash
# round 2, GPU call 4: full bench line with the new legs, row-geometry knob, LServe prefetch A/B, PMC traffic of the four GEMVs
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/c4_bench.log 2>&1
echo "== OMNI_DECODE_RT=256" > gpurun_out/c4_ab.log
OMNI_DECODE_RT=256 timeout 300 python tools/decode_ab.py --budgets 40 --policies 0 >> gpurun_out/c4_ab.log 2>&1
echo "== OMNI_DECODE_RT=512" >> gpurun_out/c4_ab.log
timeout 300 python tools/decode_ab.py --budgets 40 --policies 0 >> gpurun_out/c4_ab.log 2>&1
echo "== g128 bs64 (planner: K split for 64-row tiles)" >> gpurun_out/c4_ab.log
timeout 300 python tools/decode_ab.py --group-size 128 --batch 64 --budgets 24 --policies 0 --steps 24 >> gpurun_out/c4_ab.log 2>&1
for mb in 0 40; do echo "== lserve OMNI_PREFETCH_MB=$mb" >> gpurun_out/c4_lserve.log; OMNI_PREFETCH_MB=$mb timeout 300 python tools/lserve_steps.py >> gpurun_out/c4_lserve.log 2>&1; done
for shape in "28672 4096 16 0" "6144 4096 16 0" "4096 4096 16 1" "4096 14336 16 1"; do
  echo "== gemv $shape" >> gpurun_out/c4_pmc.log
  tools/gpu_pmc_traffic.sh w4a8_gemv python $R/tools/gemv_loop.py $shape >> gpurun_out/c4_pmc.log 2>&1
done
tail -3 gpurun_out/c4_bench.log | cut -c1-3000; cat gpurun_out/c4_ab.log gpurun_out/c4_lserve.log gpurun_out/c4_pmc.log
