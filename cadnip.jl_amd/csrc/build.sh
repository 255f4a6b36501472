#!/bin/bash
# Build libcadnip_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.  Incremental and parallel (Makefile).
set -euo pipefail
cd "$(dirname "$0")"
# the Verilog-A modules compiled into the library: one generated stamp function each (cadnip.jl_amd/va/hipgen.py)
VA_LIST=$(python3 -c "import sys; sys.path.insert(0, '../..'); import importlib; print(' '.join(importlib.import_module('cadnip_jl_amd.va').MODEL_FILES))")
(cd ../.. && python3 -m cadnip_jl_amd.va.hipgen $(for f in $VA_LIST; do echo cadnip.jl_amd/va/models/$f; done)) > va_generated.hpp.tmp
cmp -s va_generated.hpp.tmp va_generated.hpp || mv va_generated.hpp.tmp va_generated.hpp; rm -f va_generated.hpp.tmp
# `build.sh --trace` builds the diagnostic library (cycle timeline of one wave, devices.hpp CADNIP_TRACE_POINT)
if [ "${1:-}" = "--trace" ]; then
  make -j"$(nproc)" OUT=../libcadnip_hip_trace.so OBJ=_obj_trace EXTRA=-DCADNIP_TRACE
  echo "built $(realpath ../libcadnip_hip_trace.so)"
else
  make -j"$(nproc)"
  echo "built $(realpath ../libcadnip_hip.so)"
fi
