#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/attn_variants_long.py 65536 131072 262144 > gpurun_out/attn_variants_long.log 2>&1; echo "rc=$?" >> gpurun_out/attn_variants_long.log
tail -20 gpurun_out/attn_variants_long.log
