import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
import cadnip_jl_amd as cj
from cadnip_jl_amd import api, benchmarks as bm
from cadnip_jl_amd.structure import expand_breakpoints
mc = api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0})
nv = int(sys.argv[1]) if len(sys.argv)>1 else 4
nt = int(sys.argv[3]) if len(sys.argv)>3 else nv
pts = list(api.ProductSweep(vdd=np.linspace(4.5,5.5,nv), temp=np.linspace(-40,125,nt)))
sim = api.BatchSimulator(mc, pts)
st=sim.st
sim.analyze()
u0, conv, dcs = sim.dc(abstol=1e-9, mode="tranop")
print("dc conv", conv.sum(), "of", len(pts), "iters", dcs["newton_iters"])
qi=st.index_of("Q")
print("Q dc:", np.round(u0[:,qi],3))
breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
sim.h.set_spec(mode="tran")
atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
ts=np.array([150e-9,250e-9,450e-9,550e-9,700e-9])
t0=time.time()
out, per, stats = sim.h.tran_run(0.0, 7e-7, atol, 1e-4, breaks=breaks, save_t=ts, obs=[qi], fused=int(sys.argv[2]) if len(sys.argv)>2 else 0)
print("wall", time.time()-t0, stats)
t,hh,o = sim.h.tran_state()
bad = np.where(per[:,3]!=1)[0]
print("failed:", bad, [pts[i] for i in bad], t[bad], hh[bad], per[bad])
print("newton per inst min/max", per[:,0].min(), per[:,0].max(), "steps", per[:,1].min(), per[:,1].max())
print(np.round(out[:,:,0],3)[:8])
print("newton per inst: mean %.1f  p50 %.0f  p90 %.0f  max %d  -> max/mean %.3f" % (per[:,0].mean(), np.percentile(per[:,0],50), np.percentile(per[:,0],90), per[:,0].max(), per[:,0].max()/per[:,0].mean()))
w = per[:,0].reshape(-1, 8) if len(pts) % 8 == 0 else None
if w is not None:
    print("per-workgroup (8 consecutive instances) max: mean over WGs %.1f, overall max %d; efficiency of static assignment = mean/max = %.3f" % (w.max(axis=1).mean(), w.max(), per[:,0].mean()/w.max()))
