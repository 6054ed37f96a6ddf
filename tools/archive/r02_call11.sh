#!/bin/bash
set -u
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
rm -f gpurun_out/c11.log
(cd /tmp && timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|\b(TCP|TCC|TA|TD|SQ|GRBM)_[A-Za-z0-9_]+" | grep -oE "(TCP|TCC|TA|TD)_[A-Za-z0-9_]+" | sort -u | tr '\n' ' ' > $R/gpurun_out/c11_counters.txt)
tools/gpu_pmc_mem.sh w4a8_gemv python $R/tools/gemv_loop.py 7168 8192 128 0 >> gpurun_out/c11.log 2>&1
echo "== M=16 gate_up 8B for comparison" >> gpurun_out/c11.log
tools/gpu_pmc_mem.sh w4a8_gemv python $R/tools/gemv_loop.py 28672 4096 16 0 >> gpurun_out/c11.log 2>&1
cat gpurun_out/c11.log; wc -c gpurun_out/c11_counters.txt
