# dense-core size of the fused linear solve (CADNIP_F2_NC) against throughput: checks the cost model of f2_program.cpp
for nc in 0 8 12 16; do
CADNIP_F2_NC=$nc timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nc', $nc, round(d['value']/1e6,2), 'M/s', d['roofline']['avg_launch_us'])" || exit 1
done
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('auto', round(d['value']/1e6,2), 'M/s', d['roofline']['avg_launch_us'])"
