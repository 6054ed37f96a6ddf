#!/bin/bash
# Generic two-pass PMC collection: tools/gpu_pmc_generic.sh <kernel-name-substring> <command...>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
PAT=$1; shift
rm -rf $O/pmc_g1 $O/pmc_g2
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_g1 -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -3
(cd /tmp && timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_g2 -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -3
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for tag in ("pmc_g1", "pmc_g2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""):
                continue
            k = (row.get("Grid_Size"), row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        print("==", f)
        for k in sorted(agg):
            n, v = agg[k]
            print("grid %-10s %-28s dispatches %3d mean %16.1f" % (k[0], k[1], n, v / n))
PY
