for i in 1 2; do
for lib in libcadnip_hip_prev.so libcadnip_hip.so; do
CADNIP_HIP_LIB=$PWD/cadnip.jl_amd/$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value']/1e6, d['roofline']['avg_launch_us'])" || exit 1
done; done
