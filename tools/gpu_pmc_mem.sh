#!/bin/bash
# Memory-path PMC pass: tools/gpu_pmc_mem.sh <kernel-name-substring> <command...>   (one group per pass)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
PAT=$1; shift
rm -rf $O/pmc_m1 $O/pmc_m2 $O/pmc_m3
(cd /tmp && timeout 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum --kernel-trace --output-format csv -d $O/pmc_m1 -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -3
(cd /tmp && timeout 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum --kernel-trace --output-format csv -d $O/pmc_m2 -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -3
(cd /tmp && timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m3 -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -3
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for tag in ("pmc_m1", "pmc_m2", "pmc_m3"):
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
            print("grid %-10s %-32s dispatches %3d mean %16.1f" % (k[0], k[1], n, v / n))
PY
