#!/bin/bash
# Build libcadnip_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcadnip_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
hipcc $FLAGS -shared -o $OUT kernels.hip api.hip driver.hip fused.hip fused2.hip symbolic.cpp "$@"
echo "built $(realpath $OUT)"
