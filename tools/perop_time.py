"""Per-op path of the benchmark flip-flop at B instances: iterations/s of one transient through the kernels behind the callback ABI
(bench.py --fused 0) and the average durations the library's own profile reports per kernel class."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cadnip_jl_amd import api, benchmarks as bm, sweep_shard
from cadnip_jl_amd.structure import expand_breakpoints
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
pts, _ = sweep_shard.rank_points(B, 0, 1)
sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), pts)
st = sim.st
sim.analyze()
u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
sim.h.set_spec(mode="tran")
atol = st.state_abstol(**bench.ABSTOL)
for k in range(2):
    t0 = time.perf_counter()
    sim.dc(abstol=1e-9, mode="tranop", fused=True); sim.h.set_spec(mode="tran")
    out, per, stats = sim.h.tran_run(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, bench.RELTOL, breaks=expand_breakpoints(st.breakpoints, bm.DFF_TSPAN),
                                     save_t=np.linspace(*bm.DFF_TSPAN, 71), obs=[st.index_of("Q")], fused=0, newton_mode=1)
    dt = time.perf_counter() - t0
print("B = %d per-op: %.2f M it/s (%d iterations in %.3f s, %d failed)" % (B, stats["newton_iters"] / dt / 1e6, stats["newton_iters"], dt, stats["n_failed"]))
sim.close()
