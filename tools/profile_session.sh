#!/bin/bash
# One profiling session on the GPU box: the full bench line (it runs its own rocprofv3 --pmc child passes: SQ counters, then
# FETCH_SIZE and WRITE_SIZE in separate runs with the calibration copy), the rocprofv3 --kernel-trace --stats summary of the
# same command, an LDS-unit counter pass, the per-op path's kernel stats, and the wave timeline of the fused kernel;
# condensed into profiles/<tag>_*.
#   usage (through gpurun):  bash tools/profile_session.sh r02
# Outputs land in gpurun_out/profiles_<tag>/ (the only directory gpurun merges back); copy them into profiles/ afterwards.
# Counters are collected in their own passes, with --kernel-trace only (see the round brief on rocprofv3 --pmc).
set -eo pipefail
TAG=${1:?tag}
ROOT=$PWD
OUT=$ROOT/gpurun_out
P=$OUT/profiles_$TAG
mkdir -p $OUT $P
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench line done"
B=$(python3 -c "import json;print(json.load(open('$OUT/${TAG}_bench.json'))['config']['instances_per_gpu'])")
LEAN="--no-live-pmc --no-cpu-baseline --no-extras --total-instances 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o run --output-format csv -- python3 bench.py $LEAN > $OUT/${TAG}_stats.log 2>&1
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $P/${TAG}_fused_B${B}_kernel_stats.csv
echo "kernel stats done"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/${TAG}_lds -o run --output-format csv -- python3 bench.py $LEAN > $OUT/${TAG}_lds.log 2>&1
python3 tools/pmc_summary.py $OUT/${TAG}_lds > $OUT/${TAG}_lds.json
python3 tools/make_pmc_summary.py $P/${TAG}_fused_B$B $OUT/${TAG}_bench.json $OUT/${TAG}_lds.json
echo "pmc summary done"
# the per-op path (the kernels behind the callback ABI) at B = 1024 and 8192: kernel stats of the same bench command with --fused 0
for NB in 1024 8192; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_perop_${NB} -o run --output-format csv -- python3 bench.py $LEAN --fused 0 --instances $NB --steps 1 --warmup 0 > $OUT/${TAG}_perop_${NB}.log 2>&1
  cp $(find $OUT/${TAG}_perop_${NB} -name "*kernel_stats.csv" | head -1) $P/${TAG}_perop_B${NB}_kernel_stats.csv
done
echo "per-op stats done"
if [ -f cadnip.jl_amd/libcadnip_hip_trace.so ]; then
  timeout -k 10 200 python3 tools/trace_fused2.py $B > $P/${TAG}_fused_B${B}_wave_trace.txt 2>&1 || true
  timeout -k 10 200 python3 tools/trace_stamp.py 8192 > $P/${TAG}_stamp_B8192_phase_trace.txt 2>&1 || true
fi
# ---- round 3: the team kernel (few instances), the callback ABI in one call, config 5 with the step LU
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_team -o run --output-format csv -- python3 tools/single_instance.py 1 f > $P/${TAG}_single_instance.txt 2>&1 || true
cp $(find $OUT/${TAG}_team -name "*kernel_stats.csv" | head -1) $P/${TAG}_team_B1_kernel_stats.csv || true
timeout -k 10 200 python3 tools/single_instance.py 1 f > $P/${TAG}_single_instance.txt 2>&1 || true
CADNIP_F2_TEAM=0 timeout -k 10 200 python3 tools/single_instance.py 1 f >> $P/${TAG}_single_instance.txt 2>&1 || true
timeout -k 10 200 python3 tools/team_phases.py > $P/${TAG}_team_phases.txt 2>&1 || true
timeout -k 10 400 python3 tools/team_scan.py > $P/${TAG}_team_scan.txt 2>&1 || true
timeout -k 10 200 python3 tools/callback_time.py > $P/${TAG}_callback_time.txt 2>&1 || true
if [ -x tools/ubench/lat ]; then timeout -k 5 60 tools/ubench/lat > $P/${TAG}_ubench_lat.txt 2>&1 || true; timeout -k 5 60 tools/ubench/divcheck > $P/${TAG}_ubench_divcheck.txt 2>&1 || true; fi
if [ -f cadnip.jl_amd/libcadnip_hip_trace.so ]; then
  for wv in 0 1 2 3; do timeout -k 10 120 python3 tools/trace_fused2.py 1 1 $wv; done > $P/${TAG}_team_B1_wave_trace.txt 2>&1 || true
fi
for NB in 1 512; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ring_${NB} -o run --output-format csv -- python3 tools/psp103_ring.py --tspan 2e-8 --batch $NB > $OUT/${TAG}_ring_${NB}.log 2>&1 || true
  cp $(find $OUT/${TAG}_ring_${NB} -name "*kernel_stats.csv" | head -1) $P/${TAG}_psp103_ring_B${NB}_kernel_stats.csv || true
done
timeout -k 10 400 python3 tools/psp103_ring.py --tspan 1e-6 --batch 1 > $P/${TAG}_psp103_ring.txt 2>&1 || true
timeout -k 10 300 python3 tools/psp103_ring.py --tspan 2e-8 --batch 1,64,512 >> $P/${TAG}_psp103_ring.txt 2>&1 || true
timeout -k 10 300 bash tools/pmc_ring.sh > $P/${TAG}_psp103_stamp_pmc.txt 2>&1 || true
echo "round-3 legs done"
cp $OUT/${TAG}_bench.json $P/${TAG}_fused_B${B}_bench.json
tail -c 600 $P/${TAG}_fused_B${B}_bench.json; echo
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_lds $OUT/${TAG}_perop_1024 $OUT/${TAG}_perop_8192 $OUT/${TAG}_team $OUT/${TAG}_ring_1 $OUT/${TAG}_ring_512   # raw traces: large, already condensed
echo "session $TAG done"
