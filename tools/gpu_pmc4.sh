#!/bin/bash
# Four separate PMC passes (SQ issue / wait, SQ LDS + MFMA, TA + TCP, TCC) over one command, aggregated per (kernel substring, grid):
#   tools/gpu_pmc4.sh <kernel-name-substring> <command...>     (commands with ABSOLUTE script paths: rocprofv3 runs from /tmp)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
PAT=$1; shift
rm -rf $O/pmc4_*
run() { tag=$1; shift; ctrs=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc4_$tag -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -2; }
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" "$@"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "$@"
run ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "$@"
run tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "$@"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "$@"
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for tag in ("sq1", "sq2", "ta", "tcp", "tcc"):
    for f in glob.glob("gpurun_out/pmc4_%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""):
                continue
            k = (row.get("Grid_Size"), row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        for k in sorted(agg):
            n, v = agg[k]
            print("%-4s grid %-9s %-36s n %3d mean %16.1f" % (tag, k[0], k[1], n, v / n))
PY
