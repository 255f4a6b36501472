#!/bin/bash
# Register / spill / LDS numbers of the kernels of one translation unit, from the compiler's own metadata (no GPU needed):
#   tools/kernel_regs.sh fused_team.hip [extra hipcc flags]      (assembly is left in /tmp/isa/<name>-hip-amdgcn-amd-amdhsa-gfx950.s)
set -euo pipefail
src=$1; shift
csrc="$(cd "$(dirname "$0")/../cadnip.jl_amd/csrc" && pwd)"
mkdir -p /tmp/isa && cd /tmp/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm "$@" -save-temps -c -o /tmp/isa/out.o "$csrc/$src" 2>/dev/null
s=$(basename "${src%.*}")-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^\s*\.name:/ {name=$2} /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vs=$2} /\.private_segment_fixed_size:/ {sc=$2} /\.group_segment_fixed_size:/ {lds=$2} /\.wavefront_size:/ {printf "%-90s vgpr %3s (spill %3s) sgpr %3s (spill %3s) scratch %5s lds %6s\n", substr(name,1,90), vg, vs, sg, ss, sc, lds}' "$s"
