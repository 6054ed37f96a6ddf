#!/bin/bash
# Register / scratch / LDS usage of the kernels in one TU whose mangled name matches a pattern (hipcc remarks, compact).
# usage: tools/kernel_regs.sh qgemm_chn.hip midm [extra hipcc flags]
cd "$(dirname "$0")/../omniserve_amd/csrc" || exit 1
src=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -I../../include "$@" -c "$src" -o /tmp/kr_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | awk -v pat="$pat" '
  /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name); keep = (name ~ pat) }
  keep && /VGPRs:/ && !/Spill/ { v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v) }
  keep && /ScratchSize/ { sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc) }
  keep && /VGPRs Spill/ { sp=$0; sub(/.*: /,"",sp); sub(/ \[.*/,"",sp) }
  keep && /LDS Size/ { l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-70s vgpr %s scratch %s spill %s lds %s\n", name, v, sc, sp, l }'
rm -f /tmp/kr_$$.o
