"""ISCAS-85 c6288 (16 x 16 multiplier, 10 112 MOSFETs) through the deck reader and the per-op GPU path: one large circuit
instead of a batch of small ones (n = 75 908 unknowns with the level-1 cards used here; the reference benchmarks it with
PSP103: n = 212 228, rebuild 0.48 s, KLU 2.72 s, doc/c6288_bottleneck_findings.md:75-84).  The gate-level netlist is the
reference's benchmarks/vacask/c6288/cedarsim/multiplier.inc, kept as a data fixture (tests/golden/c6288_multiplier.inc).

Measures deck -> structure -> symbolic LU -> restamp -> refactor + solve, and checks the GPU solve against SciPy's SuperLU.
The DC operating point of the whole multiplier does not converge with the PCNR / gshunt / source-stepping chain -- nor does
a 12-stage inverter chain in the oracle's restatement of that loop.      python tools/c6288.py   (needs a GPU)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cadnip_jl_amd as cj
from cadnip_jl_amd import api


def deck(a=0xFFFF, b=0xFFFF, vdd=1.2):
    inc = open(os.path.join(ROOT, "tests", "golden", "c6288_multiplier.inc")).read()
    d = '* c6288\n.include "multiplier.inc"\nvdd vdd 0 %g\nvss vss 0 0\nx1 ' % vdd
    d += " ".join("a%d" % k for k in range(16)) + " " + " ".join("b%d" % k for k in range(16)) + " " + " ".join("p%d" % k for k in range(32)) + " c6288\n"
    for k in range(16):
        d += "va%d a%d 0 DC %g\nvb%d b%d 0 DC %g\n" % (k, k, vdd * ((a >> k) & 1), k, k, vdd * ((b >> k) & 1))
    nm = dict(type=1, vto=0.4, kp=200e-6, gamma=0.3, phi=0.7, cbd=1e-15, cbs=1e-15)
    pm = dict(type=-1, vto=-0.4, kp=100e-6, gamma=0.3, phi=0.7, cbd=1e-15, cbs=1e-15)
    return cj.netlist.read_spice(d, models={"nmos": nm, "pmos": pm}, includes={"multiplier.inc": inc})[0]


def main():
    import scipy.sparse as sp, scipy.sparse.linalg as spl
    t0 = time.time()
    circ = deck(0xBEEF, 0x1234)
    print("deck            %8.2f s   %d devices" % (time.time() - t0, len(circ.devices))); t0 = time.time()
    sim = api.BatchSimulator(api.MNACircuit(circ, {}, api.MNASpec(mode="tran")), [{}])
    st, h = sim.st, sim.h
    print("structure       %8.2f s   n = %d, nnz = %d (%d charges, %d limit unknowns)" % (time.time() - t0, st.n, st.nnz, st.n_charges, st.n_limits)); t0 = time.time()
    rng = np.random.default_rng(1)
    u = rng.random(st.n) * 1.2
    h.rebuild(u, 0.0)
    J = h.jacobian(1e9)[0]
    h.analyze_values(J)
    print("symbolic LU     %8.2f s   %s" % (time.time() - t0, h.lu_stats()))
    for name, fn in (("restamp", lambda: h.rebuild(u, 0.0)), ("J = G + gamma C", lambda: h.jacobian(1e9, readback=False)), ("refactor", h.factor)):
        fn()
        t0 = time.time()
        for _ in range(5):
            fn()
        print("%-15s %8.2f ms per call (host call to completion, incl. the PCIe copies of the callback ABI)" % (name, (time.time() - t0) / 5 * 1e3))
    G, C, b, _ = h.get_GCb()
    rhs = rng.random(st.n)
    t0 = time.time()
    x = h.solve(rhs)[0]
    t_solve = time.time() - t0
    A = sp.csc_matrix((G[0] + 1e9 * C[0], st.ref_rowval, st.ref_colptr), shape=(st.n, st.n))
    t0 = time.time()
    xr = spl.splu(A).solve(rhs)
    t_ref = time.time() - t0
    res = lambda v: np.linalg.norm(A @ v - rhs) / np.linalg.norm(rhs)
    print("solve           %8.2f ms;  SciPy SuperLU factor + solve on the host: %.2f s" % (t_solve * 1e3, t_ref))
    print("                relative residual |A x - b| / |b|: GPU %.2e, SuperLU %.2e;  max |x - x_ref| / max |x_ref| = %.2e (cond ~ 1e9 * C / gmin)"
          % (res(x), res(xr), np.max(np.abs(x - xr)) / np.max(np.abs(xr))))
    sim.close()


if __name__ == "__main__":
    main()
