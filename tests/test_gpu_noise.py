"""noise! through the library (src/noise.jl:118-190): DC operating point and linearisation on the GPU, sources collected on
the host at that point, the reference's adjoint sweep on the host.  Fixtures: test/noise.jl -- resistor divider (:31-46), RC low-pass (:48-66),
input-referred noise (:82-114), and -- with the oracle's registered sources travelling as data, the models' sources being absent on the GPU
box -- the sp_diode shot-noise ratio (:161-176) and the sp_bjt mechanisms (:178-189)."""
import os

import numpy as np
import pytest

from cadnip_jl_amd import api, netlist, structure as S

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KT = api.K_BOLTZMANN * (27.0 + 273.15)
KINDS = ("thermal", "shot", "white", "flicker")


def test_divider_and_rc_closed_forms():
    circ, _ = netlist.read_spice("* divider\nV1 in 0 DC 0\nR1 in out 1k\nR2 out 0 1k\n")
    ns = api.noise(api.MNACircuit(circ, {}), "out", [1.0, 1e3, 1e6], input="V1")
    assert np.allclose(ns["onoise"], 4 * KT * 500.0, rtol=1e-6) and np.allclose(ns["r1"], ns["r2"]) and np.allclose(ns["r1"] + ns["r2"], ns["onoise"])
    assert np.allclose(ns.gain.real, 0.5, rtol=1e-6) and np.allclose(ns.gain.imag, 0.0, atol=1e-9)
    assert np.allclose(ns["inoise"], 4 * KT * 500.0 / 0.25, rtol=1e-6)
    circ, _ = netlist.read_spice("* rc\nV1 in 0 DC 0\nR1 in out 1k\nC1 out 0 1u\n")
    freqs = api.acdec(10, 1.0, 1e7)
    ns = api.noise(api.MNACircuit(circ, {}), "out", freqs, input="V1")
    assert np.allclose(ns["onoise"], 4 * KT * 1e3 / (1 + (2 * np.pi * freqs * 1e3 * 1e-6) ** 2), rtol=1e-6) and np.allclose(ns["r1"], ns["onoise"])
    assert np.allclose(ns["inoise"], 4 * KT * 1e3, rtol=1e-6)
    assert api.total_noise(ns, referred="input") ** 2 == pytest.approx(4 * KT * 1e3 * (freqs[-1] - freqs[0]), rel=1e-6)
    with pytest.raises(ValueError):
        api.noise(api.MNACircuit(circ, {}), "out", [])
    with pytest.raises(KeyError):
        api.noise(api.MNACircuit(circ, {}), "out", [1e3], input="R1")


def test_builtin_diode_shot_noise():
    """devices.jl:1393-1397: the built-in diode registers 2 q |I0| at its junction bias; against R1's 4kT/R through the same impedance"""
    circ, _ = netlist.read_spice("* d\n.model dm d is=1e-14\nV1 in 0 DC 5\nR1 in out 10k\nD1 out 0 dm\n")
    mc = api.MNACircuit(circ, {})
    ns = api.noise(mc, "out", [1e2, 1e4])
    vout = api.dc(mc)["out"]
    I_D = (5.0 - vout) / 10e3
    assert I_D > 1e-4 and np.allclose(ns["d1"] / ns["r1"], 2 * api.Q_ELEMENTARY * I_D / (4 * KT / 10e3), rtol=1e-6)
    assert np.allclose(ns["d1"] + ns["r1"], ns["onoise"])


def test_simple_mosfet_and_diode_flicker_noise_through_the_library():
    """devices.jl:1718-1732 (SimpleMOSFET: channel thermal 4kT (2/3) gm, flicker KF |Ids|^AF / f^FFE) and :1435-1443 (diode flicker) at the
    GPU's operating point: the common-source stage's closed form S_out = (4kT / Rd + 4kT (2/3) gm + KF Ids^AF / f^FFE) (Rd || 1 / gds)^2."""
    from cadnip_jl_amd.circuit import Circuit
    c = Circuit("SimpleMOSFET common-source stage")
    c.V("vdd", "vdd", "0", dc=5.0)
    c.V("vg", "in", "0", dc=1.0)
    c.R("rd", "vdd", "out", 10e3)
    c.SMOS("m1", "out", "in", "0", Vth=0.5, K=1e-3, lambda_=0.02, KF=1e-14, AF=1.2, FFE=0.9)
    mc = api.MNACircuit(c, {})
    freqs = np.array([1.0, 10.0, 1e3, 1e5])
    ns = api.noise(mc, "out", freqs)
    vds = api.dc(mc)["out"]
    ids = 1e-3 / 2 * 0.5 ** 2 * (1 + 0.02 * vds)
    gm, gds = 1e-3 * 0.5 * (1 + 0.02 * vds), 1e-3 / 2 * 0.5 ** 2 * 0.02
    rout = 1.0 / (1.0 / 10e3 + gds)
    assert np.allclose(ns["onoise"], (4 * KT / 10e3 + 4 * KT * (2.0 / 3.0) * gm + 1e-14 * ids ** 1.2 / freqs ** 0.9) * rout ** 2, rtol=1e-6)
    assert np.allclose(ns["m1"] + ns["rd"], ns["onoise"])
    d = Circuit("diode flicker")
    d.V("v1", "in", "0", dc=5.0)
    d.R("r1", "in", "out", 10e3)
    d.D("d1", "out", "0", Is=1e-14, KF=1e-15, AF=1.5, FFE=1.1)
    mc = api.MNACircuit(d, {})
    nd = api.noise(mc, "out", freqs)
    i0 = (5.0 - api.dc(mc)["out"]) / 10e3
    assert np.allclose(nd["d1"] / nd["r1"], (2 * api.Q_ELEMENTARY * i0 + 1e-15 * i0 ** 1.5 / freqs ** 1.1) / (4 * KT / 10e3), rtol=1e-5)


def _fixture_noise(name):
    st, x = S.load_structure(os.path.join(GOLD, "va_%s.npz" % name))
    packed = [x["packed%d" % i] for i in range(int(x["n_packed"][0]))]
    sim = api.BatchSimulator.from_packed(st, packed, api.MNASpec(mode="dcop", temp=27.0), vscale=2.0)
    import scipy.sparse as sp
    try:
        u, conv, _ = sim.dc(abstol=1e-10, mode="dcop")
        assert conv[0]
        sim.h.rebuild(u, 0.0)
        G, C, _, _ = sim.h.get_GCb()
    finally:
        sim.close()
    dense = lambda nz: sp.csc_matrix((nz, st.ref_rowval, st.ref_colptr), shape=(st.n, st.n)).toarray()
    Gd, Cd = dense(G[0]), dense(C[0])
    Gd[np.arange(st.n_nodes), np.arange(st.n_nodes)] += 1e-12
    names = bytes(x["noise_names"]).decode().split(",")
    srcs = [(int(p) - 1, int(n) - 1, KINDS[int(k)], float(a), float(b), nm)
            for p, n, k, a, b, nm in zip(x["noise_p"], x["noise_n"], x["noise_kind"], x["noise_a"], x["noise_b"], names)]
    ns = api.noise_solve(st, Gd, Cd, srcs, bytes(x["noise_output"]).decode(), x["noise_freqs"], None, 27.0)
    return st, x, u[0], ns


def test_sp_diode_shot_noise_ratio():
    """test/noise.jl:161-176 with the GPU's operating point and linearisation"""
    st, x, u, ns = _fixture_noise("noise_diode")
    assert np.allclose(ns["onoise"], x["noise_onoise"], rtol=1e-6)
    I_D = (5.0 - u[st.index_of("out")]) / 10e3
    assert I_D > 1e-4 and np.allclose(ns["xd1_id"] / ns["r1"], 2 * api.Q_ELEMENTARY * I_D / (4 * KT / 10e3), rtol=1e-4)
    assert np.allclose(sum(ns.contributions.values()), ns["onoise"])


def test_sp_bjt_noise_mechanisms():
    """test/noise.jl:178-189"""
    st, x, u, ns = _fixture_noise("noise_bjt")
    assert np.allclose(ns["onoise"], x["noise_onoise"], rtol=1e-6)
    for mech in ("xq1_rc", "xq1_rb", "xq1_re", "xq1_ic", "xq1_ib", "xq1_flicker"):
        assert mech in ns.contributions
    assert ns["xq1_flicker"][0] / ns["xq1_flicker"][1] == pytest.approx(10.0, rel=1e-6)
    assert ns["xq1_ic"][0] == pytest.approx(ns["xq1_ic"][1], rel=1e-6)
    assert np.allclose(sum(ns.contributions.values()), ns["onoise"]) and np.all(ns["onoise"] > 0)
