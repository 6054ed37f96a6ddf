#!/bin/bash
# per-kernel table of the LServe context stage at 65 536 tokens (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT"
tools/gpu_prof_cmd.sh lserve_ctx python $GRAFT_REPO_ROOT/tools/lserve_prefill.py kv8 65536 > gpurun_out/lserve_ctx_prof.log 2>&1
head -40 gpurun_out/prof_lserve_ctx_by_grid.md
