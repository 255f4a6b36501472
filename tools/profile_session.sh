#!/bin/bash
# One profiling session on the GPU box: bench line, rocprofv3 kernel stats, PMC passes (HBM bytes, SQ wave states, LDS
# unit) and the wave timeline of the fused kernel; condensed into profiles/<tag>_*.
#   usage (through gpurun):  bash tools/profile_session.sh r01f
# Counters are collected in their own passes, with --kernel-trace only (see the round brief on rocprofv3 --pmc).
set -eo pipefail
TAG=${1:?tag}
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT $ROOT/profiles
export TMPDIR=/tmp
BENCH="bench.py --steps 1 --warmup 0 --no-cpu-baseline"
# a short bench line first: make_pmc_summary.py takes the iteration count of a step from it
timeout -k 10 200 python3 $BENCH > $OUT/${TAG}_bench1.json 2> $OUT/${TAG}_bench1.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run --output-format csv -- python3 $BENCH > $OUT/${TAG}_stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch -o run --output-format csv -- python3 $BENCH --calib-copy 1024 > $OUT/${TAG}_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write -o run --output-format csv -- python3 $BENCH --calib-copy 1024 > $OUT/${TAG}_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/${TAG}_sq -o run --output-format csv -- python3 $BENCH > $OUT/${TAG}_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/${TAG}_lds -o run --output-format csv -- python3 $BENCH > $OUT/${TAG}_lds.log 2>&1
python3 tools/pmc_summary.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_sq > $OUT/${TAG}_pmc.json
python3 tools/pmc_summary.py $OUT/${TAG}_lds > $OUT/${TAG}_lds.json
B=$(python3 -c "import json;print(json.load(open('$OUT/${TAG}_bench1.json'))['config']['instances_per_gpu'])")
python3 tools/make_pmc_summary.py ${TAG}_fused_B$B $OUT/${TAG}_bench1.json $OUT/${TAG}_pmc.json $OUT/${TAG}_lds.json > /dev/null
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}_fused_B${B}_kernel_stats.csv
cp profiles/${TAG}_fused_B${B}_pmc_summary.json profiles/${TAG}_fused_B${B}_kernel_stats.csv $OUT/
if [ -f cadnip.jl_amd/libcadnip_hip_trace.so ]; then
  timeout -k 10 200 python3 tools/trace_fused2.py $B > $OUT/${TAG}_fused_B${B}_wave_trace.txt 2>&1
fi
# the full default bench line last: it reads `traffic` from the summary written above (bench.py PMC_SUMMARY)
timeout -k 10 600 python3 bench.py > $OUT/${TAG}_fused_B${B}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_fused_B${B}_bench.json; echo
echo "session $TAG done"
