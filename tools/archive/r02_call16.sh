#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/c16.log
for n in PQB1 PIPE3 PIPE5 PIPE8; do
  echo "== $n" >> gpurun_out/c16.log
  OMNI_TUNE_LIB=tune_libs/lib_attn_$n.so timeout 300 python tools/attn_prefill_bench.py >> gpurun_out/c16.log 2>&1
done
grep -v amdgpu.ids gpurun_out/c16.log
