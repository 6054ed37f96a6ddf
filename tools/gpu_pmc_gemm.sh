#!/bin/bash
# PMC pass over the prefill W4A8 GEMM (tools/gemm4096.py): MFMA busy, LDS conflicts, wait breakdown.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
rm -rf $O/pmc_gemm $O/pmc_gemm2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_gemm -o g -- python $R/tools/gemm4096.py) 2>&1 | grep -v amdgpu.ids | tail -2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_gemm2 -o g -- python $R/tools/gemm4096.py) 2>&1 | grep -v amdgpu.ids | tail -2
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_gemm", "pmc_gemm2"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "w4a8_gemm_kernel" not in row.get("Kernel_Name", ""):
                continue
            k = (row.get("Grid_Size"), row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        print("==", f)
        for k in sorted(agg):
            n, v = agg[k]
            print("grid %-10s %-28s dispatches %3d mean %16.1f" % (k[0], k[1], n, v / n))
PY
