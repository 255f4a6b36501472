import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from cadnip_jl_amd import api, benchmarks as bm, sweep_shard
from cadnip_jl_amd.structure import expand_breakpoints
pts = sweep_shard.rank_points(4096, 0, 1)[0]
sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), pts)
st = sim.st; sim.analyze()
atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6); breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
save_t = np.linspace(0, 7e-7, 71)
for rep in range(4):
    t0=time.perf_counter(); u0, conv, dcs = sim.dc(abstol=1e-9, mode="tranop", fused=True); t1=time.perf_counter()
    sim.h.set_spec(mode="tran"); t2=time.perf_counter()
    out, per, stats = sim.h.tran_run(0.0, 7e-7, atol, 1e-4, breaks=breaks, save_t=save_t, obs=[st.index_of("Q")], fused=2, newton_mode=1); t3=time.perf_counter()
    print("dc %.2f ms  set_spec %.2f  tran_run %.2f ms (driver loop %.2f ms)  total %.2f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, stats["wall_seconds"]*1e3, (t3-t0)*1e3))
sim.close()
