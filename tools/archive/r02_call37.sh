#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_gemm_gpu.py tests/test_kv4_gpu.py tests/test_reference_layer_golden_gpu.py tests/test_persistent_gpu.py -q -x > gpurun_out/mirror_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/mirror_tests.log; tail -4 gpurun_out/mirror_tests.log
timeout 600 python bench.py --no-lserve > gpurun_out/bench_mirror.log 2>&1; python - <<'PY'
import json
for line in open("gpurun_out/bench_mirror.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print("value", d["value"], "drop_in", d.get("drop_in"), "protocol", d.get("protocol", {}).get("tokens_per_s"))
PY
