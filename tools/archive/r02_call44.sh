#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$PWD/gpurun_out
for v in old new; do
  rm -rf $O/pmc_w8_$v
  if [ $v = old ]; then export OMNI_TUNE_LIB=$PWD/tune_libs/lib_w8old.so; else unset OMNI_TUNE_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_w8_$v -o g -- python $GRAFT_REPO_ROOT/tools/gemm_w8.py) 2>&1 | grep int8_tops
done
python - <<'PY'
import csv, glob, collections
for v in ("old", "new"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/pmc_w8_%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "w4a8_gemm_kernel" not in row["Kernel_Name"] or row["Grid_Size"] != "3670016":
                continue
            agg[row["Counter_Name"]][0] += 1; agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
    print(v, {k: round(a[1] / a[0] / 1e6, 1) for k, a in sorted(agg.items())})
PY
