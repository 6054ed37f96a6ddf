#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/c20.log
for n in NOEXP NOSUM; do
  echo "== $n" >> gpurun_out/c20.log
  OMNI_TUNE_LIB=tune_libs/lib_attn_$n.so timeout 300 python tools/attn_prefill_bench.py >> gpurun_out/c20.log 2>&1
done
echo "== shipped" >> gpurun_out/c20.log
timeout 300 python tools/attn_prefill_bench.py >> gpurun_out/c20.log 2>&1
grep -v amdgpu.ids gpurun_out/c20.log
