"""Cycle split of one sp_mos1 stamping wave (csrc/stamp_csr.hip, diagnostic build `build.sh --trace`):
python tools/trace_stamp.py [n_instances]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CADNIP_HIP_LIB"] = os.path.join(ROOT, "cadnip.jl_amd", "libcadnip_hip_trace.so")
import numpy as np
from cadnip_jl_amd import api, benchmarks as bm, hip, sweep_shard

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pts, _ = sweep_shard.rank_points(B, 0, 1)
sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), pts)
sim.analyze()
u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
lib = hip.load_library()
s8 = (C.c_ulonglong * 8)(); cnt = C.c_ulonglong()
lib.cadnip_debug_stamp_trace(s8, C.byref(cnt), 1)
sim.h.set_spec(mode="tran")
for _ in range(50):
    sim.h.rebuild(u0, 1e-8)
lib.cadnip_debug_stamp_trace(s8, C.byref(cnt), 0)
n = max(cnt.value, 1)
print("B=%d waves=%d  cycles per wave (mean over all waves): prologue+zero %.0f  stamp %.0f | (unused %.0f %.0f %.0f %.0f %.0f)  reduce %.0f"
      % ((B, n) + tuple(s8[k] / n for k in range(8))))
sim.close()
