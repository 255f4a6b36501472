"""BASELINE.json config 5 on the GPU: PSP103 (the reference's models/PSPModels.jl/va/psp103.va through this build's Verilog-A
generator, csrc/va_generated_ext.hpp) -- stamps against the oracle's fixtures, the reference's two DC tests, and the 9-stage
ring oscillator of benchmarks/vacask/ring (n = 371, doc/ring_oscillator_investigation.md:22).

The model's source is not on the GPU box (third-party text inside the reference, never copied): structure, packed parameters
and the oracle's G / C / b come from tests/golden/psp103_*.npz, written by tools/make_psp103_fixtures.py where the source is at
hand (tests/test_psp103_cpu.py checks there that the committed fixtures are what the script produces today)."""
import os

import numpy as np
import pytest

from cadnip_jl_amd import api, structure as S
from cadnip_jl_amd.structure import expand_breakpoints

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-11     # stamped values relative to the largest entry of their array: PSP103 evaluates ~3 000 statements of chained
                 # transcendentals per device; the oracle (Python floats, math.*) and the GPU (ocml) differ in the last bits of
                 # exp / log / pow, which the model amplifies a little


def _load(name):
    st, x = S.load_structure(os.path.join(GOLD, "%s%s.npz" % ("" if name.startswith("bsim4") else "psp103_", name)))
    packed = [x["packed%d" % i] for i in range(int(x["n_packed"][0]))]
    return st, x, packed, bytes(x["mode"]).decode()


def _sim(name, mode=None, B=1):
    st, x, packed, fmode = _load(name)
    if B > 1:
        packed = [np.repeat(p, B, axis=0) for p in packed]
    sim = api.BatchSimulator.from_packed(st, packed, api.MNASpec(mode=mode or fmode, temp=27.0), vscale=1.2)
    return st, x, sim


@pytest.mark.parametrize("name", ["nmos_defaults", "nmos_card", "ring", "bsim4_nmos", "bsim4_dff"])
def test_psp103_stamps_match_the_oracle(name):
    """(bsim4_*: the reference's bsim4v8.va -- 13 nodes + 18 $limit sites = 31 derivative directions on 32 lanes per device)"""
    st, x, sim = _sim(name)
    h = sim.h
    worst = 0.0
    for k in range(len(x["U"])):
        h.set_initjct(False)
        h.rebuild(x["U"][k], float(x["T"][k]))
        G, C, b, _ = h.get_GCb()
        for got, ref in ((G[0], x["G"][k]), (C[0], x["C"][k]), (b[0], x["b"][k])):
            err = np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300)
            worst = max(worst, err)
            assert err <= RTOL, (name, k, err)
    sim.close()
    print(name, "worst relative stamp error", worst)


@pytest.mark.parametrize("name,lo,hi", [("nmos_defaults", 100e-6, 1e-3), ("nmos_card", 10e-6, 10e-3)])
def test_psp103_dc_tests_of_the_reference(name, lo, hi):
    """test/mna/psp103_integration.jl:40-122: V(d) = 1.2, V(g) = 0.6 to 1e-6 and the drain current in the stated window; plus
    the oracle's own DC solution of the same circuit (the PCNR Newton of oracle/mna_ref.py on the interpreted model)."""
    st, x, sim = _sim(name, mode="dcop")
    u, conv, stats = sim.dc(abstol=1e-10, mode="dcop")
    sim.close()
    assert conv[0], stats
    assert abs(u[0, st.index_of("d")] - 1.2) < 1e-6 and abs(u[0, st.index_of("g")] - 0.6) < 1e-6
    Id = u[0, st.index_of("I_Vds")]
    assert lo < abs(Id) < hi, Id
    ref = x["dc_x"]
    assert abs(Id - ref[st.index_of("I_Vds")]) <= 1e-9 * abs(Id)
    nn = st.n_nodes
    assert np.max(np.abs(u[0, :nn] - ref[:nn])) < 1e-9


def test_psp103_ring_oscillates():
    """benchmarks/vacask/ring/cedarsim/runme.jl:47-69: CedarTranOp start, dtmax = 50 ps; here the first 100 ns (the reference's
    investigation slices the run the same way, doc/ring_oscillator_investigation.md:228): 371 unknowns, every stage swings
    rail to rail after the 10 uA kick at 1 ns, all nine stages at one frequency."""
    st, x, sim = _sim("ring", mode="tran")
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges) == (371, 154, 127, 90)
    sim.analyze()
    u, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
    assert conv[0]
    assert abs(u[0, st.index_of("vdd")] - 1.2) < 1e-9
    t1 = 100e-9
    ts = np.linspace(0.0, t1, 4001)
    obs = [st.index_of(str(k)) for k in range(1, 10)]
    sim.h.set_spec(mode="tran")
    out, per, stats = sim.h.tran_run(0.0, t1, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-3,
                                     breaks=expand_breakpoints(st.breakpoints, (0.0, t1)), save_t=ts, obs=obs, hmax=50e-12, fused=0)
    sim.close()
    assert stats["n_failed"] == 0, stats
    v = out[0]
    late = ts > 40e-9
    periods = []
    for k in range(9):
        w = v[late, k]
        assert w.max() > 1.0 and w.min() < 0.2, (k, w.min(), w.max())
        up = np.flatnonzero((w[:-1] < 0.6) & (w[1:] >= 0.6))
        assert len(up) >= 3
        periods.append(np.mean(np.diff(ts[late][up])))
    assert max(periods) / min(periods) < 1.02, periods
    print("ring period %.3f ns, %d Newton iterations, %d accepted / %d rejected steps" % (
        np.mean(periods) * 1e9, stats["newton_iters"], stats["steps_accepted"], stats["steps_rejected"]))


def test_bsim4v8_dc_matches_the_oracle():
    """One sp_bsim4v8 NMOS on its default card (W = 1 u, L = 0.5 u, Vgs = 0.8, Vds = 1.2): the GPU's PCNR DC solve -- nine $limit
    unknowns, two branch currents of executed V(a,b) <+ 0 statements, two surviving internal nodes -- lands on the oracle's solution."""
    st, x, sim = _sim("bsim4_nmos", mode="dcop")
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits) == (20, 4, 4, 3, 9)
    u, conv, stats = sim.dc(abstol=1e-10, mode="dcop")
    sim.close()
    assert conv[0], stats
    ref = x["dc_x"]
    Id = u[0, st.index_of("I_Vds")]
    assert 50e-6 < abs(Id) < 500e-6 and abs(Id - ref[st.index_of("I_Vds")]) <= 1e-8 * abs(Id)
    assert np.max(np.abs(u[0, :st.n_nodes] - ref[:st.n_nodes])) < 1e-9


def test_dff_with_the_reference_bsim4v8_text_latches():
    """SURVEY.md section 8d's secondary model for config 3: the gf180 flip-flop netlist with every MOSFET an sp_bsim4v8 instance on the
    default card, Vdd = 1.8 V (n = 535: 78 nodes, 67 branch currents, 120 charge states, 270 limit unknowns).  DC start and the
    0-700 ns transient on the per-op path; the logic pins that do not depend on the D / CLKN race (test/gf180_dff.jl:29-33): Q = 0 at
    150 and 250 ns, Q = Vdd at 700 ns, Q_neg its complement."""
    st, x, sim = _sim("bsim4_dff", mode="tran")
    sim.analyze()
    u, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
    assert conv[0]
    ts = np.array([150e-9, 250e-9, 700e-9])
    sim.h.set_spec(mode="tran")
    out, per, stats = sim.h.tran_run(0.0, 7e-7, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4,
                                     breaks=expand_breakpoints(st.breakpoints, (0.0, 7e-7)), save_t=ts, obs=[st.index_of("Q"), st.index_of("Q_neg")], fused=0)
    sim.close()
    assert stats["n_failed"] == 0, stats
    q, qn = out[0, :, 0], out[0, :, 1]
    assert abs(q[0]) < 0.02 and abs(q[1]) < 0.02 and abs(q[2] - 1.8) < 0.02, q
    assert abs(qn[2]) < 0.02 and abs(qn[0] - 1.8) < 0.02, qn
    print("bsim4v8 DFF: %d Newton iterations, %d accepted / %d rejected steps, %.2f s" % (stats["newton_iters"], stats["steps_accepted"], stats["steps_rejected"], stats["wall_seconds"]))


@pytest.mark.parametrize("name,case", [("psp103_ring", "ring"), ("bsim4_dff", "bsim4_dff")])
def test_generated_model_transient_follows_the_oracle_trajectory(name, case):
    """Config 5's trajectory pin (and the bsim4v8 flip-flop's): tests/golden/<name>_tran.npz holds slices of the transient as the C++ port's
    controller and LU produce them on stamps of the literal Python interpreter (tools/make_tran_fixtures.py: oracle/va_ref.py behind
    cpu_port.Port.set_stamper; newton_mode 2 = the per-op GPU path's mode 1).  Same start state, same pivot order (chosen on the fixture's
    sample matrix), same options: the GPU's per-op path must take the same path -- identical Newton, accepted and rejected step counts -- and
    land on the same waveforms.  The bar is 1e-9 relative wherever the circuit does not amplify differences itself: all of the flip-flop's
    slice; the ring up to the kick, and the ring's second slice (a restart on the limit cycle).  Between the kick and saturation the ring
    leaves a metastable point and multiplies whatever the two sides differ by -- the stamps' last bits, 3e-14 -- by its start-up gain
    (measured: 2e-13 at 0.7 ns, 2e-10 at 1.3 ns, 3e-7 from 2 ns on, then flat): there the bar is 2e-6.  (Between this repository's
    restatements: the reference holds no waveform for these decks.)"""
    path = os.path.join(GOLD, "%s_tran.npz" % name)
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated" % path)
    f = np.load(path)
    st, x, sim = _sim(case, mode="tran")
    ref_order = np.empty(st.nnz)
    ref_order[np.asarray(st.to_ref_nz)] = f["sample"]
    sim.h.analyze_values(ref_order)
    sim.h.set_spec(mode="tran")
    slices = [(f["u0"], 0.0, float(f["t1"][0]), f["breaks"], f["save_t"], f["out"], f["counts"])]
    if "u1" in f.files:
        slices.append((f["u1"], float(f["t1"][0]), float(f["t2"][0]), f["breaks2"], f["save_t2"], f["out2"], f["counts2"]))
    for k, (u0, t0, t1, breaks, save_t, ref, counts) in enumerate(slices):
        sim.h.set_u(u0[None, :])
        out, per, stats = sim.h.tran_run(t0, t1, f["atol"], float(f["reltol"][0]), breaks=breaks, save_t=save_t, obs=[int(j) for j in f["obs"]],
                                         hmax=float(f["hmax"][0]), fused=0, newton_mode=1)
        assert stats["n_failed"] == 0, stats
        dev = np.max(np.abs(out[0] - ref) / np.maximum(np.abs(ref), 1.0), axis=1)
        print("%s slice %d: GPU %d Newton iterations, %d accepted / %d rejected; oracle %s; worst relative deviation %.2e" % (
            name, k, per[0, 0], per[0, 1], per[0, 2], counts[:3].tolist(), dev.max()))
        assert (int(per[0, 0]), int(per[0, 1]), int(per[0, 2])) == tuple(int(c) for c in counts[:3])
        if name == "psp103_ring" and k == 0:
            assert np.all(dev[save_t <= 1.0e-9] <= 1e-9) and dev.max() <= 2e-6, dev
        else:
            assert dev.max() <= 1e-9, dev
    sim.close()
