#!/bin/bash
# Shader clock under a kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration, per kernel whose name
# contains PAT:  tools/gpu_pmc_clock.sh PAT command...   (ABSOLUTE script paths: rocprofv3 runs from /tmp)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
PAT=$1; shift
rm -rf $O/pmcclk
(cd /tmp && timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmcclk -o g -- "$@") 2>&1 | grep -v "amdgpu.ids\|simple_timer" | tail -2
PAT=$PAT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
dur = {}
for f in glob.glob("gpurun_out/pmcclk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in glob.glob("gpurun_out/pmcclk/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r.get("Kernel_Name", "") and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            agg[r["Grid_Size"]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
    for g, v in sorted(agg.items()):
        v = v[len(v) // 3:]
        c = sum(x for x, _ in v) / len(v); d = sum(y for _, y in v) / len(v)
        print("grid %-9s dispatches %3d  GUI_ACTIVE %12.0f  duration %8.2f us  -> %.3f GHz (/8 XCDs)" % (g, len(v), c, d, c / 8 / d / 1e3))
PY
