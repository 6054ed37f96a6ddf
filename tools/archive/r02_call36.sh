#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/lserve_prefill.py kv8 65536 256000 > gpurun_out/lserve_prefill.log 2>&1; echo "rc=$?" >> gpurun_out/lserve_prefill.log
tail -4 gpurun_out/lserve_prefill.log
timeout 600 python tools/lserve_prefill.py kv4 256000 > gpurun_out/lserve_prefill_kv4.log 2>&1; echo "rc=$?" >> gpurun_out/lserve_prefill_kv4.log
tail -3 gpurun_out/lserve_prefill_kv4.log
