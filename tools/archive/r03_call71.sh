#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
tools/gpu_prof_cmd.sh r03_j_lserve python $R/tools/lserve_steps.py kv8 24 > gpurun_out/r03_j_lserve_prof.log 2>&1
tools/gpu_prof_cmd.sh r03_j_tp python $R/tools/tp_rank_steps.py 128 > gpurun_out/r03_j_tp_prof.log 2>&1
grep -v "at::native\|rocprim" gpurun_out/prof_r03_j_lserve_by_grid.md | head -16 | cut -c1-170
