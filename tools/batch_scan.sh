# throughput against the sweep size (instances per GPU): how much of a step is the tail where the in-kernel queue is empty
for b in 1024 2048 4096 8192 16384; do
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --instances $b 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($b, round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms', d['config'].get('launches_last_step'))" || exit 1
done
