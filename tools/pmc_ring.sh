cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_ring_$tag -o run --output-format csv -- python3 tools/psp103_ring.py --tspan 2e-9 --batch 1 > gpurun_out/pmc_ring_$tag.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py gpurun_out/pmc_ring_SQ_INSTS_VALU gpurun_out/pmc_ring_SQ_INSTS_BRANCH > gpurun_out/pmc_ring_summary.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_ring_summary.json'))
for k,v in d.items():
    if 'PSP103' in k and 'stamp' in k:
        print(k[:80])
        for c,x in sorted(v.items()): print("  %-24s %12.1f per call" % (c, x['per_call']))
PY
rm -rf gpurun_out/pmc_ring_SQ_INSTS_VALU gpurun_out/pmc_ring_SQ_INSTS_BRANCH
