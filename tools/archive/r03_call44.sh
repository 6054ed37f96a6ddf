#!/bin/bash
# PMC of the long-context decode attention: round-3 head (two tiles per batch, old softmax) vs the committed kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c44; O=gpurun_out/r3c44; rm -f $O/*.log
OMNI_TUNE_LIB=tune_libs/libhead.so bash tools/gpu_pmc_attn.sh kv4 > $O/pmc_head.log 2>&1
bash tools/gpu_pmc_attn.sh kv4 > $O/pmc_new.log 2>&1
paste -d'|' <(grep "grid" $O/pmc_head.log | cut -c1-90) <(grep "grid" $O/pmc_new.log | cut -c40-90)
