#!/bin/bash
# Build a tuning variant of the library: tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" [CSRC_DIR]
# -> tune_libs/libNAME.so (git-ignored, travels with gpurun).  Used with OMNI_TUNE_LIB=tune_libs/libNAME.so tools/*.py.
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=$2; SRC=${3:-omniserve_amd/csrc}
OBJ=/tmp/variant_$NAME; mkdir -p $OBJ tune_libs
pids=()
for f in qgemm_plan qgemm_chn qgemm_grp qgemm_w8 elementwise offpath kv_cache attn_prefill sparse_utils tp_comm row_dtypes norm_gemv_fused; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $EXTRA -I$SRC -Iinclude -c $SRC/$f.hip -o $OBJ/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o tune_libs/lib$NAME.so
echo built tune_libs/lib$NAME.so
