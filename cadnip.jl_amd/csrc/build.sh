#!/bin/bash
# Build libcadnip_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.  Incremental and parallel (Makefile).
set -euo pipefail
cd "$(dirname "$0")"
# the Verilog-A modules compiled into the library: one generated stamp function each (cadnip.jl_amd/va/hipgen.py)
VA_LIST=$(python3 -c "import sys; sys.path.insert(0, '../..'); import importlib; print(' '.join(importlib.import_module('cadnip_jl_amd.va').MODEL_FILES))")
(cd ../.. && python3 -m cadnip_jl_amd.va.hipgen $(for f in $VA_LIST; do echo cadnip.jl_amd/va/models/$f; done)) > va_generated.hpp.tmp
cmp -s va_generated.hpp.tmp va_generated.hpp || mv va_generated.hpp.tmp va_generated.hpp; rm -f va_generated.hpp.tmp
# the external models (the reference's own model files: sources inside the reference checkout or $CADNIP_VA_PATH, never copied): the
# generated table (va_generated_ext.hpp) and translation units (va_ext/<module>.hip) are committed and only regenerated when every
# source is found
EXT_LIST=$(python3 -c "import sys; sys.path.insert(0, '../..'); import importlib; va = importlib.import_module('cadnip_jl_amd.va'); ps = [va.external_source(fn, sd) for _, fn, sd in va.EXTERNAL]; print(' '.join(ps) if all(ps) else '')")
if [ -n "$EXT_LIST" ]; then
  (cd ../.. && python3 -m cadnip_jl_amd.va.hipgen --ext cadnip.jl_amd/csrc/va_ext $EXT_LIST) > va_generated_ext.hpp.tmp
  cmp -s va_generated_ext.hpp.tmp va_generated_ext.hpp || mv va_generated_ext.hpp.tmp va_generated_ext.hpp; rm -f va_generated_ext.hpp.tmp
fi
# `build.sh --trace` builds the diagnostic library (cycle timeline of one wave, devices.hpp CADNIP_TRACE_POINT)
if [ "${1:-}" = "--trace" ]; then
  # --trace [1|2]: 1 = phase boundaries only, 2 (default) = every point (devices.hpp)
  make -j"$(nproc)" OUT=../libcadnip_hip_trace.so OBJ=_obj_trace${2:-2} EXTRA=-DCADNIP_TRACE=${2:-2}
  echo "built $(realpath ../libcadnip_hip_trace.so)"
else
  make -j"$(nproc)"
  echo "built $(realpath ../libcadnip_hip.so)"
fi
