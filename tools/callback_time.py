"""The callback sequence a Julia integrator would drive -- cadnip_rebuild -> residual -> jacobian -> factor -> solve with host pointers -- timed
at B = 1 on the benchmark flip-flop (bench.py reports the same as callback_us_per_iter)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cadnip_jl_amd import api, benchmarks as bm
sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), None)
st, h = sim.st, sim.h
sim.analyze()
u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
h.set_spec(mode="tran")
du = np.zeros((1, st.n)); g = np.array([1e9])
def seq():
    h.rebuild(u0, 5e-8); r = h.residual(du, u0); h.jacobian(g, download=False) if "download" in h.jacobian.__code__.co_varnames else h.jacobian(1e9); h.factor(); return h.solve(r)
for _ in range(20): x = seq()
t0 = time.perf_counter(); N = 200
for _ in range(N): x = seq()
print("callback sequence: %.1f us per iteration; |x| = %.6e" % (1e6 * (time.perf_counter() - t0) / N, float(np.abs(x).sum())))
sim.close()
