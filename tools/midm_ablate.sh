#!/bin/bash
# gate_up at M = 128 / 64 under the timing ablations of w4a8_midm_kernel (tune_libs/libmidm_ab*.so; WRONG results by design)
cd "$(dirname "$0")/.."
for v in ${ABL:-0 1 2 4 8 16 31}; do
  if [ $v = 0 ]; then lib=""; else lib="tune_libs/libmidm_ab$v.so"; fi
  echo "== ablate $v"
  OMNI_TUNE_LIB=$lib python tools/midm_sweep.py --one 2>&1 | grep "midm" | sed -e 's/  */ /g' | cut -c1-100
done
