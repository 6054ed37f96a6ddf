#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== pytest kv4" ; timeout 900 python -m pytest tests/test_kv4_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider 2>&1 | tail -30 | tee $O/pytest_kv4.log
echo "== attn sweep" ; timeout 600 python - <<'PY' 2>&1 | tee $O/attn_sweep.log
import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
import tools.sweep_graph as sg
for kern in (-2, -1):
    sg.lib.omni_kv4_decode_set_split_override(kern)
    print("kernel", "mfma" if kern == -2 else "valu")
    sg.sweep_attn(16, 1024)
sg.lib.omni_kv4_decode_set_split_override(-2)
sg.sweep_attn(64, 1024)
sg.sweep_attn(4, 4096)
PY
