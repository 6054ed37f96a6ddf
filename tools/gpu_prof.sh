#!/bin/bash
# rocprofv3 kernel stats of the bench decode loop + separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the
# gate_up GEMV.  Outputs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=$PWD/gpurun_out
R=$PWD
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 32 --warmup 4 --no-extras) 2>&1 | tail -3
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o gemv -- python $R/tools/gemv_loop.py) 2>&1 | tail -2
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o gemv -- python $R/tools/gemv_loop.py) 2>&1 | tail -2
find $O/prof $O/pmc_fetch $O/pmc_write -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True):
    print("==", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 24: print(",".join(row))
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        rd = csv.DictReader(open(f))
        for row in rd:
            k = (row.get("Kernel_Name", "")[:60], row.get("Counter_Name"))
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        print("==", f)
        for k, (n, v) in agg.items():
            print(k, "dispatches", n, "mean", v / n)
PY
