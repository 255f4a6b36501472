import numpy as np, sys
sys.path.insert(0,'/root/repo')
import cadnip_jl_amd as cj
from cadnip_jl_amd import api, benchmarks as bm
mc = api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0})
sim = api.BatchSimulator(mc)
st=sim.st
sim.analyze()
print(sim.h.lu_stats())
u0, conv, dcs = sim.dc(abstol=1e-9, mode="tranop")
print("dc", conv, dcs["newton_iters"], [ (nm, round(u0[0][st.index_of(nm)],4)) for nm in ("Q","Q_neg","net0","net7","D_neg","cki","ncki")])
from cadnip_jl_amd.structure import expand_breakpoints
breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
print("breaks", breaks)
sim.h.set_spec(mode="tran")
atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
ts=np.linspace(0,7e-7,71)
for maxit in (50, 200, 1000, 5000, 20000):
    sim.h.set_u(u0)
    out, per, stats = sim.h.tran_run(0.0, 7e-7, atol, 1e-4, breaks=breaks, save_t=ts, max_iterations=maxit, obs=[st.index_of("Q"), st.index_of("CLKN"), st.index_of("D")])
    t,hh,o = sim.h.tran_state()
    print(maxit, "t=",t, "h=",hh, "ord",o, per, {k:stats[k] for k in ("newton_iters","steps_accepted","steps_rejected","newton_failures","launches","wall_seconds")})
    if per[0,3]!=0: break
print(out[0][:, 0])
