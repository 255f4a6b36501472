#!/bin/bash
# Build libcadnip_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcadnip_hip.so
# `build.sh --trace` builds the diagnostic library (cycle timeline of one wave, devices.hpp CADNIP_TRACE_POINT)
if [ "${1:-}" = "--trace" ]; then shift; OUT=../libcadnip_hip_trace.so; set -- -DCADNIP_TRACE "$@"; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
hipcc $FLAGS -shared -o $OUT kernels.hip api.hip driver.hip fused2.hip symbolic.cpp "$@"
echo "built $(realpath $OUT)"
