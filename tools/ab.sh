# A/B of two builds of the library within one GPU session (box-to-box variation is ~10 %): the current build against
# cadnip.jl_amd/libcadnip_hip_prev.so, which you build from the commit to compare with, e.g.
#   git archive <commit> cadnip.jl_amd/csrc include | tar -x -C /tmp/prev && /tmp/prev/cadnip.jl_amd/csrc/build.sh && cp /tmp/prev/cadnip.jl_amd/libcadnip_hip.so cadnip.jl_amd/libcadnip_hip_prev.so
for i in 1 2; do
for lib in libcadnip_hip_prev.so libcadnip_hip.so; do
CADNIP_HIP_LIB=$PWD/cadnip.jl_amd/$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value']/1e6, d['roofline']['avg_launch_us'])" || exit 1
done; done
