#!/bin/bash
# Short GPU iteration: parity tests + probes (no bench / profile).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider 2>&1 | tail -30 | tee $O/pytest.log
echo "== debug" ; timeout 300 python tools/debug_norm.py 2>&1 | tail -40 | tee $O/debug_norm.log
echo "== stream" ; (hipcc --offload-arch=gfx950 -O3 -w tools/stream_probe.hip -o /tmp/stream_probe && timeout 300 /tmp/stream_probe) 2>&1 | tail -50 | tee $O/stream_probe.log
