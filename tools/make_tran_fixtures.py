"""Trajectory fixtures for circuits of generated Verilog-A models (BASELINE.json config 5: the PSP103 ring; SURVEY 8d's secondary model: the
bsim4v8 flip-flop).  The C++ port (oracle/cpu_port.cpp) knows the built-in devices and the hand-written sp_mos1 only, so these circuits had
no transient oracle.  Here the port's transient controller and LU run on EXTERNAL stamps: every fast_rebuild! is the literal Python
interpreter (oracle/va_ref.py through oracle/mna_ref.py: fast_rebuild!) -- the same source text, the oracle's own dual numbers -- handed to
the port through a callback (cpu_port.Port.set_stamper).  Policy: the port's newton_mode 2 = what the per-op GPU path does with
newton_mode 1 (IDA's convergence test, a refactorisation every round).

Written to tests/golden/<case>_tran.npz (DATA: start state, the sample matrix the pivot order is chosen on, options, the recorded
waveform and the step / Newton counts); the structure and parameters are those of tests/golden/<case>.npz.
The reference holds no waveform for these decks (benchmarks/vacask/ring/cedarsim/runme.jl:47-69 prints timings; compare_ngspice.jl's raw
file is absent), so this pins the GPU trajectory against this repository's independent restatement, not against the reference's IDA.

    python tools/make_tran_fixtures.py [psp103_ring] [bsim4_dff]        (needs /root/reference or CADNIP_VA_PATH; minutes of CPU)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cadnip_jl_amd as cj                                   # noqa: E402
from cadnip_jl_amd import api, hip, netlist                  # noqa: E402
from cadnip_jl_amd.structure import TYPE_ID, expand_breakpoints   # noqa: E402
from oracle import cpu_port, mna_ref as M                    # noqa: E402
from oracle.netlist_ref import make_builder                  # noqa: E402
import make_psp103_fixtures as F                             # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# case -> (fixture case of make_psp103_fixtures, t1, tolerances, reltol, hmax, observed nets, voltage scale of the pivot samples, end of the restart slice or 0)
CASES = {
    "psp103_ring": ("ring", 8.0e-9, dict(vntol=1e-4, iabstol=1e-7, chgtol=1e-4), 1e-2, 50e-12, [str(k) for k in range(1, 10)], 1.2, 12.0e-9),
    "bsim4_dff": ("bsim4_dff", 5.6e-8, dict(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-3, 0.0, ["Q", "Q_neg", "CLKN", "D"], 1.8, 0.0),
}


def build(name):
    case, t1, tols, reltol, hmax, nets, vscale, t2 = CASES[name]
    deck, includes, mode = F.cases()[case]
    circ = deck if not isinstance(deck, str) else netlist.read_spice(deck, includes=includes)[0]
    st = cj.discover(circ, {})
    packed = cj.pack_params(st, circ, {}, np.array([27.0]), 1, gmin=1e-12)
    bld = make_builder(circ.to_dicts({}))
    spec = M.MNASpec(mode="tran", temp=27.0)
    ctx = M.build_with_detection(bld, {}, spec)
    cs = M.compile_structure(bld, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    assert np.array_equal(st.ref_colptr, cs.colptr) and np.array_equal(st.ref_rowval, cs.rowval)
    to_ref = np.asarray(st.to_ref_nz)
    n_rebuild = [0]

    def stamper(u, t):
        n_rebuild[0] += 1
        M.fast_rebuild(ws, u, t)
        lw = np.zeros(st.n)
        if st.n_limits:
            lw[st.n - st.n_limits:] = ws.dctx.limit_w
        return cs.G.data[to_ref], cs.C.data[to_ref], ws.dctx.b, lw

    # start state: the oracle's own CedarTranOp solution
    sol = M.solve_dc(bld, {}, M.MNASpec(mode="tranop", temp=27.0))
    assert sol.converged, name
    u0 = np.asarray(sol.x, dtype=float)
    port = cpu_port.Port(st, [p[0] for p in packed], {**TYPE_ID, **{b.type: 99 for b in st.blocks if b.type not in TYPE_ID}})
    port.set_stamper(stamper)
    # pivot order: the product's host symbolic phase on the element-wise max |G + 1e9 C| over sample states (as BatchSimulator.analyze)
    rng = np.random.default_rng(1234)
    acc = np.zeros(st.nnz)
    for k in range(5):
        u = u0.copy() if k == 0 else np.zeros(st.n) if k == 1 else (rng.random(st.n) * 1.2 - 0.1) * vscale
        if k >= 2:
            u[st.n_nodes:st.n_nodes + st.n_currents] = 0.0
        G, C, _, _ = port.rebuild(u, 0.0)
        acc = np.maximum(acc, api.clip_sample(G + 1e9 * C)[0])
    prog = hip.host_lu_analyze(st.n, st.rowptr, st.colidx, acc, sample=True, leaves=hip.leaves_of(st))
    port.set_lu(prog)
    atol = st.state_abstol(**tols)
    breaks = expand_breakpoints(st.breakpoints, (0.0, t1))
    obs = [st.index_of(nm) for nm in nets]
    ts = np.linspace(0.0, t1, 61)
    t0 = time.time()
    out, uf, stats, _ = port.tran(u0, 0.0, t1, atol, reltol, breaks=breaks, save_t=ts, obs=obs, err_mask=st.differential_mask(), use_pcnr=False,
                                  hmax=hmax, newton_mode=2)
    assert stats["status"] == 1, stats
    print("%-12s n = %d: %d Newton iterations, %d accepted / %d rejected steps, %d oracle rebuilds in %.0f s" % (
        name, st.n, stats["newton_iters"], stats["accepted"], stats["rejected"], n_rebuild[0], time.time() - t0))
    d = dict(u0=u0, sample=acc, t1=np.array([t1]), reltol=np.array([reltol]), hmax=np.array([hmax]), atol=atol, breaks=np.asarray(breaks, dtype=float),
             save_t=ts, obs=np.array(obs), out=out, counts=np.array([stats["newton_iters"], stats["accepted"], stats["rejected"], stats["newton_failures"]]),
             rperm=np.asarray(prog["rperm"]), cperm=np.asarray(prog["cperm"]), nets=np.frombuffer(",".join(nets).encode(), dtype=np.uint8))
    if t2 > t1:
        # second slice: a restart from the state the first one ended in.  The ring leaves a metastable operating point at the kick and a start-up
        # amplifies whatever two implementations differ by (rounding of the stamps, 1e-14) by the loop gain until the oscillation saturates;
        # once it runs on its limit cycle differences no longer grow -- this slice is where a tight waveform comparison means something.
        ts2 = np.linspace(t1, t2, 41)
        out2, _, st2, _ = port.tran(uf, t1, t2, atol, reltol, breaks=expand_breakpoints(st.breakpoints, (t1, t2)), save_t=ts2, obs=obs,
                                    err_mask=st.differential_mask(), use_pcnr=False, hmax=hmax, newton_mode=2)
        assert st2["status"] == 1, st2
        print("             restart %.1f - %.1f ns: %d Newton iterations, %d accepted / %d rejected" % (t1 * 1e9, t2 * 1e9, st2["newton_iters"], st2["accepted"], st2["rejected"]))
        d.update(u1=uf, t2=np.array([t2]), save_t2=ts2, out2=out2, breaks2=np.asarray(expand_breakpoints(st.breakpoints, (t1, t2)), dtype=float),
                 counts2=np.array([st2["newton_iters"], st2["accepted"], st2["rejected"], st2["newton_failures"]]))
    port.close()
    return d


def main():
    for name in (sys.argv[1:] or list(CASES)):
        d = build(name)
        path = os.path.join(GOLD, "%s_tran.npz" % name)
        np.savez_compressed(path, **d)
        print("  ->", os.path.relpath(path, ROOT), "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
