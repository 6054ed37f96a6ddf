#!/bin/bash
# round 3, call 16: where the exact prefill kernel's time goes -- per-workgroup timeline (debug clocks) + PMC passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out; O=$PWD/gpurun_out/r3c16; mkdir -p $O; R=$PWD
OMNI_TUNE_LIB=tune_libs/libclk.so timeout 300 python tools/gemm_timeline.py > $O/timeline.log 2>&1
rm -rf $O/pmc1 $O/pmc2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -o g -- python $R/tools/gemm4096.py) 2>&1 | grep -v amdgpu.ids | tail -2
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -o g -- python $R/tools/gemm4096.py) 2>&1 | grep -v amdgpu.ids | tail -2
python - <<'PY' > $O/pmc_summary.txt
import csv, glob, collections
for tag in ("pmc1", "pmc2"):
    for f in glob.glob("gpurun_out/r3c16/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "w4a8_gemm_" not in row.get("Kernel_Name", ""):
                continue
            k = (row.get("Grid_Size"), row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        print("==", f)
        for k in sorted(agg):
            n, v = agg[k]
            print("grid %-10s %-28s dispatches %3d mean %16.1f" % (k[0], k[1], n, v / n))
PY
rm -rf $O/pmc1 $O/pmc2
cat $O/timeline.log $O/pmc_summary.txt | grep -v amdgpu.ids
