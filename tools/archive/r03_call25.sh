#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3c25
hipcc --offload-arch=gfx950 -O3 -w tools/stream_dma_probe.hip -o /tmp/sdp && timeout 120 /tmp/sdp 2>&1 | tee gpurun_out/r3c25/stream_dma.log
