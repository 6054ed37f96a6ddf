#!/bin/bash
# profile pass of the new defaults: decode step per kernel (down prefetched by norm2), PMC traffic + bench constants, LServe level 3 per kernel
cd "$(dirname "$0")/.."
R=$PWD
bash tools/r03_profile.sh r03_g
tools/gpu_prof_cmd.sh r03_g_lserve python $R/tools/lserve_steps.py kv8 24 > gpurun_out/r03_g_lserve_prof.log 2>&1
tail -5 gpurun_out/r03_g_constants.log; head -30 gpurun_out/prof_r03_g_by_grid.md | cut -c1-200; head -24 gpurun_out/prof_r03_g_lserve_by_grid.md | cut -c1-200
