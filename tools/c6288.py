"""ISCAS-85 c6288 (16 x 16 multiplier, 10 112 MOSFETs) through the deck reader and the per-op GPU path: one large circuit
instead of a batch of small ones (n = 75 908 unknowns with the level-1 cards used here; the reference benchmarks it with
PSP103: n = 212 228, rebuild 0.48 s, KLU 2.72 s, doc/c6288_bottleneck_findings.md:75-84).  The gate-level netlist is the
reference's benchmarks/vacask/c6288/cedarsim/multiplier.inc, kept as a data fixture (tests/golden/c6288_multiplier.inc).

Measures deck -> structure -> symbolic LU -> restamp -> refactor + solve, checks the GPU solve against SciPy's SuperLU, and
runs the power-up transient: supplies and inputs ramp up, the multiplier settles, the 32 output bits are the product.
(A static DC solve of the whole multiplier does not converge with the PCNR / gshunt / source-stepping chain -- nor does a
12-stage inverter chain in the oracle's restatement of that loop)      python tools/c6288.py   (needs a GPU)"""
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


def powerup(a=0xBEEF, b=0x1234, vdd=1.2, t_end=10e-9, verbose=False):
    """Supplies and inputs ramp from 0 in 1 ns, the transient starts from the all-zero state (consistent: every source is
    at 0 V) -- the start-up that a static DC solve of this circuit does not find.  Returns (product read from the output
    bits at t_end, per-instance statistics, seconds)."""
    inc = open(os.path.join(ROOT, "tests", "golden", "c6288_multiplier.inc")).read()
    ramp = "PWL(0 0 1n %g)"
    d = '* c6288 power-up\n.include "multiplier.inc"\nvdd vdd 0 ' + ramp % vdd + '\nvss vss 0 0\nx1 '
    d += " ".join("a%d" % k for k in range(16)) + " " + " ".join("b%d" % k for k in range(16)) + " " + " ".join("p%d" % k for k in range(32)) + " c6288\n"
    for k in range(16):
        d += "va%d a%d 0 %s\nvb%d b%d 0 %s\n" % (k, k, ramp % vdd if (a >> k) & 1 else "DC 0", k, k, ramp % vdd if (b >> k) & 1 else "DC 0")
    nm = dict(type=1, vto=0.4, kp=200e-6, gamma=0.3, phi=0.7, cbd=1e-15, cbs=1e-15)
    pm = dict(type=-1, vto=-0.4, kp=100e-6, gamma=0.3, phi=0.7, cbd=1e-15, cbs=1e-15)
    circ = cj.netlist.read_spice(d, models={"nmos": nm, "pmos": pm}, includes={"multiplier.inc": inc})[0]
    sim = api.BatchSimulator(api.MNACircuit(circ, {}, api.MNASpec(mode="tran")), [{}])
    st, h = sim.st, sim.h
    t0 = time.time()
    # CedarUICOp-style start (dcop.jl:109-151): no DC solve, backward-Euler steps from the zero state
    out, per, stats = sim.tran((0.0, t_end), st.state_abstol(vntol=1e-4, iabstol=1e-7, chgtol=1e-4), 1e-3, np.array([t_end]),
                               initializealg="uic", obs=[st.index_of("p%d" % k) for k in range(32)], fused=0)
    dt = time.time() - t0
    sim.close()
    p = sum((1 << k) for k in range(32) if out[0, 0, k] > 0.5 * vdd)
    rails = bool(np.all((np.abs(out[0, 0]) < 0.05 * vdd) | (np.abs(out[0, 0] - vdd) < 0.05 * vdd)))
    if verbose:
        print("power-up transient to %.0f ns: %d Newton iterations, %d steps (+%d rejected), status %d, %.1f s = %.1f ms per iteration"
              % (t_end * 1e9, per[0, 0], per[0, 1], per[0, 2], per[0, 3], dt, dt / max(1, per[0, 0]) * 1e3))
        print("product bits 0x%08X, expected 0x%04X * 0x%04X = 0x%08X: %s (all outputs at a rail: %s)" % (p, a, b, a * b, "OK" if p == a * b else "MISMATCH", rails))
    return p, per[0], dt


def powerup_batch(pairs, vdd=1.2, t_end=10e-9, verbose=False):
    """Several operand pairs at once: one sweep instance per pair (the input sources' amplitudes are sweep parameters), one
    1024-thread LU workgroup per instance.  Returns the products read back, and the seconds the transient took."""
    from cadnip_jl_amd.circuit import Param
    inc = open(os.path.join(ROOT, "tests", "golden", "c6288_multiplier.inc")).read()
    d = '* c6288 power-up, operands as sweep parameters\n.include "multiplier.inc"\nvdd vdd 0 PWL(0 0 1n %g)\nvss vss 0 0\nx1 ' % vdd
    d += " ".join("a%d" % k for k in range(16)) + " " + " ".join("b%d" % k for k in range(16)) + " " + " ".join("p%d" % k for k in range(32)) + " c6288\n"
    for k in range(16):
        d += "va%d a%d 0 PWL(0 0 1n 1)\nvb%d b%d 0 PWL(0 0 1n 1)\n" % (k, k, k, k)
    nm = dict(type=1, vto=0.4, kp=200e-6, gamma=0.3, phi=0.7, cbd=1e-15, cbs=1e-15)
    pm = dict(type=-1, vto=-0.4, kp=100e-6, gamma=0.3, phi=0.7, cbd=1e-15, cbs=1e-15)
    circ = cj.netlist.read_spice(d, models={"nmos": nm, "pmos": pm}, includes={"multiplier.inc": inc})[0]
    for dev in circ.devices:
        if dev.type == "V" and dev.name[:2] in ("va", "vb") and dev.name[2:].isdigit():
            dev.params["scale"] = Param("%s%s" % (dev.name[1], dev.name[2:]))      # amplitude of bit k of operand a / b
    pts = []
    for a, b in pairs:
        pt = {}
        for k in range(16):
            pt["a%d" % k] = vdd * ((a >> k) & 1)
            pt["b%d" % k] = vdd * ((b >> k) & 1)
        pts.append(pt)
    sim = api.BatchSimulator(api.MNACircuit(circ, dict(pts[0]), api.MNASpec(mode="tran")), pts)
    st = sim.st
    t0 = time.time()
    out, per, stats = sim.tran((0.0, t_end), st.state_abstol(vntol=1e-4, iabstol=1e-7, chgtol=1e-4), 1e-3, np.array([t_end]),
                               initializealg="uic", obs=[st.index_of("p%d" % k) for k in range(32)], fused=0)
    dt = time.time() - t0
    sim.close()
    prods = [sum((1 << k) for k in range(32) if out[i, 0, k] > 0.5 * vdd) for i in range(len(pairs))]
    if verbose:
        print("batch of %d multipliers: %.1f s, %d Newton iterations in total = %.2f ms per iteration and instance" % (
            len(pairs), dt, stats["newton_iters"], dt / max(1, stats["newton_iters"]) * 1e3))
        for (a, b), p in zip(pairs, prods):
            print("   0x%04X * 0x%04X -> 0x%08X %s" % (a, b, p, "OK" if p == a * b else "MISMATCH (expected 0x%08X)" % (a * b)))
    return prods, dt


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
    powerup(verbose=True)
    powerup_batch([(0xBEEF, 0x1234), (0xFFFF, 0xFFFF), (0x0001, 0x8000), (0xA5A5, 0x5A5A), (0x0000, 0xFFFF), (0x7FFF, 0x0003), (0x1357, 0x2468), (0xC0DE, 0xF00D)], verbose=True)


if __name__ == "__main__":
    main()
