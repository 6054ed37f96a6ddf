#!/bin/bash
# HBM traffic of the decode attention kernel (KV4 and KV8, B = 8, T = 32768) vs its algorithmic bytes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c75; O=gpurun_out/r3c75; rm -f $O/*.log
(echo "== kv4"; bash tools/gpu_pmc_traffic.sh decode_flash python $PWD/tools/attn_long.py kv4 4) > $O/pmc.log 2>&1
(echo "== kv8"; bash tools/gpu_pmc_traffic.sh decode_flash python $PWD/tools/attn_long.py kv8 4) >> $O/pmc.log 2>&1
grep -v amdgpu $O/pmc.log | cut -c1-200
