#!/bin/bash
# PMC passes over the long-context decode attention (tools/attn_long.py): issue / wait breakdown, LDS, MFMA.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
MODE=${1:-kv4}
rm -rf $O/pmc_attn1 $O/pmc_attn2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_attn1 -o g -- python $R/tools/attn_long.py $MODE 4) 2>&1 | grep -v amdgpu.ids | tail -2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_attn2 -o g -- python $R/tools/attn_long.py $MODE 4) 2>&1 | grep -v amdgpu.ids | tail -2
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_attn1", "pmc_attn2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "decode_flash" not in row.get("Kernel_Name", ""):
                continue
            k = (row.get("Grid_Size"), row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        print("==", f)
        for k in sorted(agg):
            n, v = agg[k]
            print("grid %-10s %-28s dispatches %3d mean %16.1f" % (k[0], k[1], n, v / n))
PY
