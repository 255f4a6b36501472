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
    st, x = S.load_structure(os.path.join(GOLD, "psp103_%s.npz" % name))
    packed = [x["packed%d" % i] for i in range(int(x["n_packed"][0]))]
    return st, x, packed, bytes(x["mode"]).decode()


def _sim(name, mode=None, B=1):
    st, x, packed, fmode = _load(name)
    if B > 1:
        packed = [np.repeat(p, B, axis=0) for p in packed]
    sim = api.BatchSimulator.from_packed(st, packed, api.MNASpec(mode=mode or fmode, temp=27.0), vscale=1.2)
    return st, x, sim


@pytest.mark.parametrize("name", ["nmos_defaults", "nmos_card", "ring"])
def test_psp103_stamps_match_the_oracle(name):
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
