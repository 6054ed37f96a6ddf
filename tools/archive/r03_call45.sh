#!/bin/bash
# decode attention isolated A/B (kernel_bench now honours OMNI_TUNE_LIB): round-3 head vs committed; PMC of both
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c45; O=gpurun_out/r3c45; rm -f $O/*.log
for lib in head new head new; do
  if [ $lib = new ]; then unset OMNI_TUNE_LIB; else export OMNI_TUNE_LIB=tune_libs/lib$lib.so; fi
  echo "== $lib" >> $O/kv.log
  timeout 300 python tools/kernel_bench.py kv kv8 2>&1 | grep -v amdgpu.ids | grep -i "decode" | cut -c17-130 >> $O/kv.log
done
unset OMNI_TUNE_LIB
OMNI_TUNE_LIB=tune_libs/libhead.so bash tools/gpu_pmc_attn.sh kv4 > $O/pmc_head.log 2>&1
bash tools/gpu_pmc_attn.sh kv4 > $O/pmc_new.log 2>&1
cat $O/kv.log; paste -d'|' <(grep "grid" $O/pmc_head.log | cut -c1-90) <(grep "grid" $O/pmc_new.log | cut -c40-90)
