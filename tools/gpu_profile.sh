#!/bin/bash
# profile pass (tools/gpu_call.sh TAG profile): per-kernel table of the decode step (level 3), PMC traffic of the gate_up kernel, bench constants
cd "$(dirname "$0")/.."
R=$PWD
TAG=${1:-r03_a}
mkdir -p gpurun_out
tools/gpu_prof_cmd.sh ${TAG} python $R/bench.py --steps 32 --warmup 4 --no-extras > gpurun_out/${TAG}_prof.log 2>&1
(echo "== gemv N K M form = 28672 4096 16 silu"; tools/gpu_pmc_traffic.sh w4a8_gemv_kernel python $R/tools/gemv_loop.py 28672 4096 16 silu) > gpurun_out/${TAG}_pmc.log 2>&1
python tools/make_bench_constants.py gpurun_out/prof_${TAG}_by_grid.md gpurun_out/${TAG}_pmc.log gpurun_out/bench_constants.json ${TAG} > gpurun_out/${TAG}_constants.log 2>&1
