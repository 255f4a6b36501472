"""Average duration of the sp_mos1 stamping kernel of the benchmark flip-flop at B instances (bench.py: stamp_kernel_leg).
    python tools/stamp_time.py [B]            (CADNIP_SC_PAD=<bytes>: extra LDS per workgroup, to see what occupancy is worth)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from cadnip_jl_amd import benchmarks as bm     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
print(json.dumps(bench.stamp_kernel_leg(bm.dff_circuit(), B)))
