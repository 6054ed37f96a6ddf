#!/bin/bash
# HBM traffic and clocks of the prefill attention at 16K vs 128K (dense causal, 32 heads)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=$PWD/gpurun_out
for L in 16384 131072; do
  rm -rf $O/pmc_a1_$L $O/pmc_a2_$L $O/kt_$L
  (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_a1_$L -o g -- python $GRAFT_REPO_ROOT/tools/attn_one.py $L dense) 2>&1 | tail -2
  (cd /tmp && timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_a2_$L -o g -- python $GRAFT_REPO_ROOT/tools/attn_one.py $L dense) 2>&1 | tail -2
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$L -o g -- python $GRAFT_REPO_ROOT/tools/attn_one.py $L dense) 2>&1 | tail -2
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_a*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if "prefill_attn" not in row.get("Kernel_Name", ""):
            continue
        agg[row.get("Counter_Name")][0] += 1; agg[row.get("Counter_Name")][1] += float(row.get("Counter_Value", 0))
    print("==", f)
    for k in sorted(agg):
        print("  %-28s dispatches %d mean %.4g" % (k, agg[k][0], agg[k][1] / agg[k][0]))
for f in sorted(glob.glob("gpurun_out/kt_*/**/*kernel_stats.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "prefill_attn" in row.get("Name", ""):
            print(f.split("/")[1], row.get("Name")[:40], "calls", row.get("Calls"), "avg ns", row.get("AverageNs"))
PY
