#!/bin/bash
# Per-op path (one kernel per device type + assemble + LU, the north_star's literal pipeline): kernel durations and
# HBM bytes per call from rocprofv3, condensed into profiles/<tag>_perop_B${NB}_hbm.json
#   usage (through gpurun):  bash tools/profile_perop.sh r01g
set -eo pipefail
TAG=${1:?tag}
OUT=$PWD/gpurun_out
mkdir -p $OUT profiles
export TMPDIR=/tmp
NB=${2:-1024}
BENCH="bench.py --steps 1 --warmup 0 --no-cpu-baseline --fused 0 --instances $NB"
timeout -k 10 300 python3 $BENCH > $OUT/${TAG}_perop_bench.json 2> $OUT/${TAG}_perop_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_perop_stats -o run --output-format csv -- python3 $BENCH > $OUT/${TAG}_perop_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_perop_fetch -o run --output-format csv -- python3 $BENCH --calib-copy 1024 > $OUT/${TAG}_perop_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_perop_write -o run --output-format csv -- python3 $BENCH --calib-copy 1024 > $OUT/${TAG}_perop_write.log 2>&1
python3 tools/pmc_summary.py $OUT/${TAG}_perop_fetch $OUT/${TAG}_perop_write > $OUT/${TAG}_perop_pmc.json
python3 - "$TAG" "$NB" <<'PY'
import csv, glob, json, sys
tag, nb = sys.argv[1], sys.argv[2]
out = "gpurun_out/"
pmc = json.load(open(out + tag + "_perop_pmc.json"))
cal = pmc['cadnip::k_calib_copy_f64(double const*, double*, long)']
fc = 1024 * 1024 / cal['FETCH_SIZE']['per_call']; wc = 1024 * 1024 / cal['WRITE_SIZE']['per_call']
stats = {}
for fn in glob.glob(out + tag + "_perop_stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        stats[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
bench = json.load(open(out + tag + "_perop_bench.json"))
res = {"note": "per-op path, %d instances; rocprofv3 --kernel-trace --stats durations, --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
               "(KB; FETCH_SIZE x %.3f, WRITE_SIZE x %.3f from the 1 GiB calibration copy); GB/s = measured HBM bytes per call / average duration"
               % (bench['config']['instances_per_gpu'], fc, wc),
       "bench_value_newton_iters_per_s": bench['value'], "kernels": {}}
for k, v in pmc.items():
    if 'cadnip' not in k or k not in stats or 'calib' in k:
        continue
    byt = (v['FETCH_SIZE']['per_call'] * fc + v['WRITE_SIZE']['per_call'] * wc) * 1024
    calls, avg_ns = stats[k]
    res["kernels"][k] = {"calls": calls, "avg_us": avg_ns / 1e3, "hbm_bytes_per_call": int(byt), "hbm_GB_per_s": byt / avg_ns}
json.dump(res, open("profiles/%s_perop_B%s_hbm.json" % (tag, nb), "w"), indent=1)
json.dump(res, open(out + "%s_perop_B%s_hbm.json" % (tag, nb), "w"), indent=1)
for k, r in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])[:8]:
    print("%-70s %7d calls %9.1f us %12d B %8.1f GB/s" % (k[:70], r["calls"], r["avg_us"], r["hbm_bytes_per_call"], r["hbm_GB_per_s"]))
PY
cp $(find $OUT/${TAG}_perop_stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}_perop_B${NB}_kernel_stats.csv
cp profiles/${TAG}_perop_B${NB}_kernel_stats.csv $OUT/
echo "per-op session $TAG done"
