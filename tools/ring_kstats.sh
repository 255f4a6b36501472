# kernel statistics of the PSP103 ring at B = 1 and the ring's timing at several batch sizes: bash tools/ring_kstats.sh <tag>
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_ring1 -o run --output-format csv -- python3 tools/psp103_ring.py --tspan 2e-8 --batch 1 > gpurun_out/${TAG}_ring1.log 2>&1 || exit 1
tail -1 gpurun_out/${TAG}_ring1.log
python3 - $(find gpurun_out/${TAG}_ring1 -name "*kernel_stats.csv") <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]: print("  %-90s calls %6s avg %9.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/${TAG}_ring1
timeout -k 10 300 python3 tools/psp103_ring.py --tspan 2e-8 --batch 64,512 > gpurun_out/${TAG}_ring.log 2>&1 || exit 1
cat gpurun_out/${TAG}_ring.log
