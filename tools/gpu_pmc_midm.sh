#!/bin/bash
# L2 -> L1 requests, L2 hits / misses and HBM traffic of w4a8_midm_kernel (gate_up 28672 x 4096 at M = 128), one counter group per pass,
# every pass under its own 100-s timeout (a TA-counter pass once hung for 15 minutes on this pool).
cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp; O=$PWD/gpurun_out
for C in "TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_')
  rm -rf $O/pmc_midm_$tag
  (cd /tmp && timeout 100 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_midm_$tag -o t -- python $R/tools/midm_one.py 128 28672 4096 chn 12) 2>&1 | grep -v "amdgpu.ids\|simple_timer\|output_stream\|tool.cpp" | tail -1
  python - "$O/pmc_midm_$tag" <<'PY'
import csv, glob, collections, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if "midm" not in row.get("Kernel_Name", ""): continue
        k = row.get("Counter_Name"); agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
    for k, (n, v) in sorted(agg.items()): print("%-24s dispatches %3d mean %16.1f" % (k, n, v / n))
PY
done
