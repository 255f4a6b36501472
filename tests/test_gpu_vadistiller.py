"""test/mna/vadistiller_integration.jl "Tier 6: Full VADistiller Models (from file)" on the GPU: every model of the reference's
models/VADistillerModels.jl that this build's Verilog-A generator compiles into the library (cadnip.jl_amd/va: EXTERNAL ->
csrc/va_ext/<module>.hip, one stamping kernel per model) in the reference's own small DC circuit.

Per circuit: (1) the stamped G, C, b at five probe states equal the oracle's (oracle/va_ref.py interpreting the model text on the
oracle's own duals, through oracle/mna_ref.py's literal fast_rebuild!), (2) the GPU's DC solve lands on the oracle's solution and
inside the window the reference's test asserts.  The model sources are not on the GPU box: structure, packed parameters and the oracle's
numbers come from tests/golden/vad_*.npz (tools/make_psp103_fixtures.py; tests/test_psp103_cpu.py keeps them current)."""
import os

import numpy as np
import pytest

from cadnip_jl_amd import api, hip, structure as S

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-11
# case -> (probe net, lower, upper): the bounds of the reference's test (vadistiller_integration.jl, line ranges in the fixture tool)
TIER6 = {"inductor": ("mid", -0.01, 0.01), "vdmos": ("drain", 0.0, 10.0), "resistor": ("mid", 2.5 - 1e-9, 2.5 + 1e-9), "capacitor": ("mid", 5.0 - 1e-6, 5.0 + 1e-6), "diode": ("diode_a", 0.6, 0.7),
         "diode_rs": ("diode_a", 0.6, 0.71), "bjt": ("collector", 0.0, 5.0), "jfet1": ("drain", 0.0, 10.0), "mes1": ("drain", 0.0, 5.0),
         "jfet2": ("drain", 0.0, 10.0), "mos1": ("drain", 0.0, 5.0), "mos2": ("drain", 0.0, 5.0), "mos3": ("drain", 0.0, 5.0),
         "mos6": ("drain", 0.0, 5.0), "mos9": ("drain", 0.0, 5.0), "bsim3v3": ("drain", 0.0, 1.8), "bsim4v8": ("drain", 0.9, 1.0)}


OTHER = ("bsimcmg_nmos", "juncap200", "nlvcr", "tm_1d", "tm_1d_interior", "tm_2d")   # models/CMCModels.jl (BSIM-CMG), JUNCAP200 of models/PSPModels.jl: pinned by the oracle's interpreter alone;
                                                 # test/NLVCR.va: ddx() with respect to a branch potential (test/ddx.jl); tm_*: test/mna/fixtures/table_model


def _sim(name):
    st, x = S.load_structure(os.path.join(GOLD, "%s_%s.npz" % ("va" if name in OTHER else "vad", name)))
    packed = [x["packed%d" % i] for i in range(int(x["n_packed"][0]))]
    return st, x, api.BatchSimulator.from_packed(st, packed, api.MNASpec(mode="dcop", temp=27.0), vscale=2.0)


@pytest.mark.parametrize("name", sorted(TIER6) + list(OTHER))
def test_stamps_match_the_oracle(name):
    st, x, sim = _sim(name)
    h = sim.h
    for k in range(len(x["U"])):
        h.set_initjct(False)
        try:
            h.rebuild(x["U"][k], float(x["T"][k]))
        except hip.CadnipError as e:         # CADNIP_NONFINITE: legitimate only where the oracle's stamps are non-finite too
            assert e.code == hip.NONFINITE and not (np.isfinite(x["G"][k]).all() and np.isfinite(x["b"][k]).all()), (name, k)
        G, C, b, _ = h.get_GCb()
        for got, ref, nm in ((G[0], x["G"][k], "G"), (C[0], x["C"][k], "C"), (b[0], x["b"][k], "b")):
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin), (name, k, nm)      # (sp_mos3 / sp_mos9: the same non-finite partials as the oracle)
            if fin.any():
                err = np.max(np.abs(got[fin] - ref[fin])) / max(np.max(np.abs(ref[fin])), 1e-300)
                assert err <= RTOL, (name, k, nm, err)
    sim.close()


@pytest.mark.parametrize("name", sorted(TIER6))
def test_dc_solution_matches_the_oracle_and_the_reference_bounds(name):
    probe, lo, hi = TIER6[name]
    st, x, sim = _sim(name)
    ref, ok = x["dc_x"], bool(x["dc_ok"][0])
    try:
        u, conv, stats = sim.dc(abstol=1e-10, mode="dcop")
    except hip.CadnipError as e:             # the product refuses non-finite stamps loudly (cadnip_rebuild: CADNIP_NONFINITE)
        assert e.code == hip.NONFINITE and not ok, name
        conv, stats = [False], None
    sim.close()
    assert bool(conv[0]) == ok, (name, stats)
    if not ok:
        # the reference's own chain ends unconverged here (sqrt(0 * dual) in saturation: a NaN partial under ForwardDiff, and on the GPU);
        # its test asserts the window on whatever solve_dc returns, which the oracle's returned state satisfies
        assert lo < ref[st.index_of(probe)] < hi
        return
    v = u[0, st.index_of(probe)]
    assert lo < v < hi, (name, v)
    nn = st.n_nodes + st.n_currents
    scale = np.maximum(np.abs(ref[:nn]), 1e-3)
    assert np.max(np.abs(u[0, :nn] - ref[:nn]) / scale) < 1e-8, (name, u[0, :nn], ref[:nn])


def test_diode_series_resistance_keeps_its_internal_node():
    """vadistiller_integration.jl:322-346: with rs = 10 the anode's internal node exists, the junction sits at 0.6 .. 0.7 V and the drop over rs
    is positive and below 10 mV; without rs the node collapses onto the terminal and one limit unknown remains (:286-299)."""
    st, x, sim = _sim("diode_rs")
    u, conv, _ = sim.dc(abstol=1e-10, mode="dcop")
    sim.close()
    assert conv[0]
    internal = [nm for nm in st.node_names if nm.endswith("sp_diode_a_int")]
    assert len(internal) == 1
    va, vi = u[0, st.index_of("diode_a")], u[0, st.index_of(internal[0])]
    assert 0.6 < vi < 0.7 and 0.0 < va - vi < 0.01
    st0, _ = S.load_structure(os.path.join(GOLD, "vad_diode.npz"))
    assert st0.n_nodes == 2 and st0.n_limits == 1


@pytest.mark.parametrize("name", OTHER)
def test_other_model_packages_dc(name):
    """BSIM-CMG (one fin, default card, Vgs = 0.6, Vds = 0.8) and a JUNCAP200 diode behind 100 Ohm: the GPU's DC solve equals the oracle's."""
    st, x, sim = _sim(name)
    u, conv, stats = sim.dc(abstol=1e-10, mode="dcop")
    sim.close()
    assert conv[0] and bool(x["dc_ok"][0]), stats
    nn = st.n_nodes + st.n_currents
    ref = x["dc_x"]
    assert np.max(np.abs(u[0, :nn] - ref[:nn]) / np.maximum(np.abs(ref[:nn]), 1e-3)) < 1e-8
    if name == "bsimcmg_nmos":
        assert 1e-6 < abs(u[0, st.index_of("I_Vds")]) < 1e-4
    if name.startswith("tm_"):    # test/mna/table_model.jl:83-96: $table_model of the parameters, evaluated when they are packed
        want = {"tm_1d": 0.02, "tm_1d_interior": 0.015, "tm_2d": 2 * 1.55 + 3 * 25.0 + 5}[name]
        assert abs(u[0, st.index_of("I_V1")] + want) < 1e-8
    if name == "nlvcr":          # test/ddx.jl:66-73: I(d,s) = V(d,s) ddx(R V(g,s)^2, V(g,s)) = 2 R V(d,s) V(g,s) = 60 A out of V1
        assert abs(u[0, st.index_of("vcc")] - 5.0) < 1e-10 and abs(u[0, st.index_of("vg")] - 3.0) < 1e-10
        assert abs(u[0, st.index_of("I_V1")] + 60.0) < 1e-6
