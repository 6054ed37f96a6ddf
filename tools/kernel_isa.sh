#!/bin/bash
# ISA of one kernel of a TU: tools/kernel_isa.sh qgemm_chn.hip 'midm_kernelILi8ELi0ELb0ELb1' out.s [extra hipcc flags]
cd "$(dirname "$0")/../omniserve_amd/csrc" || exit 1
src=$1; pat=$2; out=$3; shift 3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -I../../include "$@" -S --cuda-device-only "$src" -o /tmp/ki_$$.s 2>/dev/null || exit 1
awk -v pat="$pat" '/^_ZN/ { keep = ($0 ~ pat) } /^\.Lfunc_end/ { if (keep) print; keep = 0 } keep' /tmp/ki_$$.s > "$out"
rm -f /tmp/ki_$$.s
wc -l "$out"
