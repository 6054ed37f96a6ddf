#!/bin/bash
# Two SQ PMC passes (issue / wait, LDS + MFMA) over one command, each under its own timeout, aggregated per grid for kernels whose
# name contains PAT:  tools/gpu_pmc_sq.sh PAT command...   (ABSOLUTE script paths: rocprofv3 runs from /tmp)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
PAT=$1; shift
rm -rf $O/pmcsq_*
run() { tag=$1; shift; ctrs=$1; shift
  (cd /tmp && timeout 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmcsq_$tag -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -2; }
run 1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS" "$@"
run 2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "$@"
run 3 "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8 SQ_WAVES SQ_INSTS_FLAT" "$@"
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
for tag in ("1", "2", "3"):
    for f in glob.glob("gpurun_out/pmcsq_%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if pat not in row.get("Kernel_Name", ""):
                continue
            k = (row.get("Grid_Size"), row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        for k in sorted(agg):
            n, v = agg[k]
            print("grid %-9s %-28s n %3d mean %14.0f" % (k[0], k[1], n, v / n))
PY
