#!/bin/bash
# Build libcadnip_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libcadnip_hip.so
# `build.sh --trace` builds the diagnostic library (cycle timeline of one wave, devices.hpp CADNIP_TRACE_POINT)
if [ "${1:-}" = "--trace" ]; then shift; OUT=../libcadnip_hip_trace.so; set -- -DCADNIP_TRACE "$@"; fi
# -disable-machine-licm: the fused kernel runs two waves per SIMD (256 VGPRs); hoisting loop-invariant constants and
# address arithmetic out of its round loop costs 80 more spilled VGPRs, each reloaded from scratch (HBM latency) at every use
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -mllvm -disable-machine-licm"
# the Verilog-A modules compiled into the library: one generated stamp function each (cadnip.jl_amd/va/hipgen.py)
VA_DIR=../va/models
VA_LIST=$(python3 -c "import sys; sys.path.insert(0, '../..'); import importlib; print(' '.join(importlib.import_module('cadnip_jl_amd.va').MODEL_FILES))")
(cd ../.. && python3 -m cadnip_jl_amd.va.hipgen $(for f in $VA_LIST; do echo cadnip.jl_amd/va/models/$f; done)) > va_generated.hpp.tmp
cmp -s va_generated.hpp.tmp va_generated.hpp || mv va_generated.hpp.tmp va_generated.hpp; rm -f va_generated.hpp.tmp
hipcc $FLAGS -shared -o $OUT kernels.hip api.hip driver.hip fused2.hip symbolic.cpp f2_program.cpp "$@"
echo "built $(realpath $OUT)"
