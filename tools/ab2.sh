# usage: bash tools/ab2.sh ENVVAR   -- same library, with and without ENVVAR=1 (one box, alternating)
for i in 1 2; do
for v in 0 1; do
if [ $v = 1 ]; then export $1=1; else unset $1; fi
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1=$v', d['value']/1e6, d['roofline']['avg_launch_us'])" || exit 1
done; done
