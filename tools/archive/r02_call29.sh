#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/lserve_prefill.py kv8 16384 65536 256000 > gpurun_out/lserve_prefill.log 2>&1; echo "rc=$?" >> gpurun_out/lserve_prefill.log
tail -20 gpurun_out/lserve_prefill.log
