# small batches: waves per workgroup (CADNIP_F2_WPB) against throughput
for b in 512 1024 1536; do for w in 8 4 2; do
CADNIP_F2_WPB=$w timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --instances $b 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($b, $w, round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms')" || exit 1
done; done
