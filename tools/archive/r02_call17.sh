#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/c17.log
for n in NODMA1 NODMA2; do
  echo "== $n" >> gpurun_out/c17.log
  OMNI_TUNE_LIB=tune_libs/lib_attn_$n.so timeout 300 python tools/attn_prefill_bench.py >> gpurun_out/c17.log 2>&1
done
# the in-tree library of this call is the PQB=2 pipelined build: run the parity suite on it
grep -v amdgpu.ids gpurun_out/c17.log
