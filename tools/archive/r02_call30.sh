#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 500 python tools/attn_balance.py 32768 131072 > gpurun_out/attn_balance.log 2>&1; echo "rc=$?" >> gpurun_out/attn_balance.log
tail -20 gpurun_out/attn_balance.log
