"""noise! (src/noise.jl) on the CPU side: the oracle's restatement (oracle/mna_ref.py: noise, the context's noise channel, the registrations of
oracle/devices_ref.py and oracle/va_ref.py) against the closed forms test/noise.jl asserts, and the product's host logic (api.noise_sources,
api.noise_solve, NoiseSol, total_noise) against the oracle at the oracle's DC point (no GPU: the DC point and G, C come from the oracle here;
tests/test_gpu_noise.py runs the same through the library)."""
import os

import numpy as np
import pytest

import cadnip_jl_amd as cj
from cadnip_jl_amd import api, netlist, va
from oracle import mna_ref as M
from oracle.netlist_ref import make_builder

KT = M.K_BOLTZMANN * (27.0 + 273.15)
DIVIDER = "* divider\nV1 in 0 DC 0\nR1 in out 1k\nR2 out 0 1k\n"
RC = "* rc\nV1 in 0 DC 0\nR1 in out 1k\nC1 out 0 1u\n"
DIODE = "* diode_shot (test/noise.jl:14-19)\nV1 in 0 DC 5\nR1 in out 10k\nXd1 out 0 sp_diode is=1e-14 rs=0\n"
BJT = "* bjt_noise (test/noise.jl:21-27)\nVcc vcc 0 DC 5\nVb b 0 DC 0.7\nRc vcc c 4.7k\nXq1 c b 0 0 sp_bjt bf=100 is=1e-15 rb=100 re=1 rc=10 kf=1e-12 af=1\n"
HAVE_SOURCE = all(va.external_source(fn, sd) is not None for _, fn, sd in va.EXTERNAL)
needs_source = pytest.mark.skipif(not HAVE_SOURCE, reason="the Verilog-A sources are not at hand")


def oracle_noise(deck, output, freqs, input=None):
    circ, _ = netlist.read_spice(deck)
    return M.noise(make_builder(circ.to_dicts({})), {}, M.MNASpec(temp=27.0), output, freqs, input)


def product_noise_at_the_oracle_point(deck, output, freqs, input=None, gmin=1e-12):
    """api.noise without the GPU: DC point, G and C from the oracle; sources, adjoint sweep and result object from the product"""
    circ = deck if not isinstance(deck, str) else netlist.read_spice(deck)[0]
    st = cj.discover(circ, {})
    bld = make_builder(circ.to_dicts({}))
    spec = M.MNASpec(mode="dcop", temp=27.0)
    sol = M.solve_dc(bld, {}, spec)
    assert sol.converged
    ctx = M.build_with_detection(bld, {}, spec)
    cs = M.compile_structure(bld, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    M.fast_rebuild(ws, np.asarray(sol.x, dtype=float), 0.0)
    G, C = cs.G.toarray(), cs.C.toarray()
    G[np.arange(st.n_nodes), np.arange(st.n_nodes)] += gmin
    srcs = api.noise_sources(st, circ, {}, np.asarray(sol.x, dtype=float), 27.0)
    return api.noise_solve(st, G, C, srcs, output, freqs, input, 27.0)


def test_resistor_divider_white_noise():
    """test/noise.jl:31-46: 4kT (R1 || R2), two equal contributions"""
    on, c, _, _ = oracle_noise(DIVIDER, "out", [1.0, 1e3, 1e6])
    assert np.allclose(on, 4 * KT * 500.0, rtol=1e-6)
    assert np.allclose(c["r1"], c["r2"]) and np.allclose(c["r1"] + c["r2"], on)
    ns = product_noise_at_the_oracle_point(DIVIDER, "out", [1.0, 1e3, 1e6])
    assert np.allclose(ns["onoise"], on, rtol=1e-12) and np.allclose(ns["r1"], c["r1"], rtol=1e-12)


def test_rc_low_pass_and_kt_over_c():
    """test/noise.jl:48-80: 4kTR / (1 + (2 pi f R C)^2); its band integral is kT/C"""
    freqs = api.acdec(10, 1.0, 1e7)
    on, c, _, _ = oracle_noise(RC, "out", freqs)
    expected = 4 * KT * 1e3 / (1 + (2 * np.pi * freqs * 1e3 * 1e-6) ** 2)
    assert np.allclose(on, expected, rtol=1e-6) and on[0] == pytest.approx(4 * KT * 1e3, rel=1e-3) and np.allclose(c["r1"], on)
    ns = product_noise_at_the_oracle_point(RC, "out", freqs)
    assert np.allclose(ns["onoise"], on, rtol=1e-12)
    dense = np.linspace(0.0, 5e6, 20001)           # (the reference integrates 200 001 points; the tail beyond the 159 Hz pole is smooth)
    nd = product_noise_at_the_oracle_point(RC, "out", dense)
    assert api.total_noise(nd) ** 2 == pytest.approx(KT / 1e-6, rel=2e-2)


def test_input_referred_noise():
    """test/noise.jl:82-114: the divider's gain is 0.5 and flat; the RC's input-referred noise is the bare 4kTR"""
    on, _, g, inn = oracle_noise(DIVIDER, "out", [1.0, 1e3, 1e6], "V1")
    assert np.allclose(g.real, 0.5, rtol=1e-6) and np.allclose(g.imag, 0.0, atol=1e-9) and np.allclose(inn, 4 * KT * 500.0 / 0.25, rtol=1e-6)
    ns = product_noise_at_the_oracle_point(DIVIDER, "out", [1.0, 1e3, 1e6], "V1")
    assert np.allclose(ns.gain, g) and np.allclose(ns["inoise"], inn, rtol=1e-12) and np.allclose(ns["inoise"], ns["onoise"] / np.abs(ns.gain) ** 2)
    freqs = api.acdec(10, 1.0, 1e7)
    nr = product_noise_at_the_oracle_point(RC, "out", freqs, "V1")
    assert np.allclose(nr["inoise"], 4 * KT * 1e3, rtol=1e-6)
    assert api.total_noise(nr, referred="input") ** 2 == pytest.approx(4 * KT * 1e3 * (freqs[-1] - freqs[0]), rel=1e-6)


def test_errors_and_edge_cases():
    """test/noise.jl:141-158"""
    with pytest.raises(ValueError):
        oracle_noise(DIVIDER, "out", [])
    with pytest.raises(ValueError):
        product_noise_at_the_oracle_point(DIVIDER, "out", [])
    ns = product_noise_at_the_oracle_point(DIVIDER, "out", [1e3])
    with pytest.raises(KeyError):
        ns["nonexistent_source"]
    with pytest.raises(KeyError):
        ns["inoise"]
    with pytest.raises(KeyError):
        api.total_noise(ns, referred="input")
    with pytest.raises(ValueError):
        api.total_noise(ns, referred="bogus")
    with pytest.raises(KeyError):
        product_noise_at_the_oracle_point(DIVIDER, "out", [1e3], "R1")      # not an independent voltage source
    with pytest.raises(KeyError):
        oracle_noise(DIVIDER, "out", [1e3], "R1")
    assert api.total_noise(ns) == pytest.approx(np.sqrt(ns["onoise"][0]))


@needs_source
def test_diode_shot_noise_against_the_resistor():
    """test/noise.jl:161-176: sp_diode's junction shot noise 2 q I_D against R1's 4kT/R -- both see the same impedance, so the ratio of their
    contributions is the ratio of their current PSDs"""
    on, c, _, _ = oracle_noise(DIODE, "out", [1e2, 1e4])
    circ, _ = netlist.read_spice(DIODE)
    sol = M.solve_dc(make_builder(circ.to_dicts({})), {}, M.MNASpec(mode="dcop", temp=27.0))
    I_D = (5.0 - sol["out"]) / 10e3
    assert I_D > 1e-4 and "xd1_id" in c
    assert np.allclose(c["xd1_id"] / c["r1"], 2 * M.Q_ELEMENTARY * I_D / (4 * KT / 10e3), rtol=1e-4)
    assert np.allclose(sum(c.values()), on)
    ns = product_noise_at_the_oracle_point(DIODE, "out", [1e2, 1e4])
    assert set(ns.contributions) == set(c)
    for k in c:
        assert np.allclose(ns[k], c[k], rtol=1e-7, atol=0.0), k


@needs_source
def test_bjt_mechanisms_and_flicker():
    """test/noise.jl:178-189: per-mechanism sources (rc, rb, re thermal; ic, ib shot; flicker rolling off as 1/f)"""
    on, c, _, _ = oracle_noise(BJT, "c", [1e1, 1e2])
    for mech in ("xq1_rc", "xq1_rb", "xq1_re", "xq1_ic", "xq1_ib", "xq1_flicker"):
        assert mech in c
    assert c["xq1_flicker"][0] / c["xq1_flicker"][1] == pytest.approx(10.0, rel=1e-6)
    assert c["xq1_ic"][0] == pytest.approx(c["xq1_ic"][1], rel=1e-6)
    assert np.allclose(sum(c.values()), on) and np.all(on > 0)
    ns = product_noise_at_the_oracle_point(BJT, "c", [1e1, 1e2])
    assert set(ns.contributions) == set(c)
    for k in c:
        assert np.allclose(ns[k], c[k], rtol=1e-6, atol=1e-40), k     # (the oracle evaluates at pnjlim(V, limit unknown), the product at V: they agree to the DC tolerance times exp sensitivity)
    assert np.allclose(ns["onoise"], on, rtol=1e-6)


@needs_source
def test_hand_written_mos1_devices_take_their_sources_from_the_model_text():
    """An inverter stage on level-1 cards: the deck's `.model nmos` MOSFET is the hand-written sp_mos1 device of this build; its noise sources
    are those of the reference's mos1.va evaluated at the solution -- the same the oracle registers when the deck instantiates the module."""
    card = "level=1 vto=0.7 kp=100u lambda=0.01 rd=50"
    hand = "* cs stage\n.model nch nmos %s\nVdd vdd 0 DC 3\nVin in 0 DC 1.2\nRd vdd out 10k\nM1 out in 0 0 nch W=20u L=1u\n" % card
    gen = "* cs stage\nVdd vdd 0 DC 3\nVin in 0 DC 1.2\nRd vdd out 10k\nXm1 out in 0 0 sp_mos1 type=1 %s w=20u l=1u\n" % card.replace("level=1 ", "")
    on, c, _, _ = oracle_noise(gen, "out", [1e2, 1e5])
    ns = product_noise_at_the_oracle_point(hand, "out", [1e2, 1e5])
    assert np.allclose(ns["onoise"], on, rtol=1e-6)
    assert sorted(k.replace("xm1", "m1") for k in c) == sorted(ns.contributions)
    for k in c:
        assert np.allclose(ns[k.replace("xm1", "m1")], c[k], rtol=1e-6, atol=1e-60), k
    assert any(np.any(v > 0) for k, v in ns.contributions.items() if k.startswith("m1_"))      # the channel's thermal noise at least


def _common_source(KF=0.0, AF=1.0, FFE=1.0):
    from cadnip_jl_amd.circuit import Circuit
    c = Circuit("SimpleMOSFET common-source stage")
    c.V("vdd", "vdd", "0", dc=5.0)
    c.V("vg", "in", "0", dc=1.0)
    c.R("rd", "vdd", "out", 10e3)
    c.SMOS("m1", "out", "in", "0", Vth=0.5, K=1e-3, lambda_=0.02, KF=KF, AF=AF, FFE=FFE)
    return c


def test_simple_mosfet_channel_thermal_and_flicker_noise():
    """The reference's SimpleMOSFET registers 4kT (2/3) gm between drain and source where it conducts (devices.jl:1718-1724,
    context.jl:1076-1077) and KF |Ids|^AF / f^FFE with KF > 0 (devices.jl:1725-1732).  Closed form of a common-source stage with a
    resistive load: S_out = (4kT / Rd + 4kT (2/3) gm + KF Ids^AF / f^FFE) (Rd || 1/gds)^2, at the oracle's operating point; the
    product's host logic gives the same numbers."""
    freqs = np.array([1.0, 10.0, 1e3, 1e5])
    for kf, af, ffe in ((0.0, 1.0, 1.0), (1e-14, 1.2, 0.9)):
        circ = _common_source(kf, af, ffe)
        bld = make_builder(circ.to_dicts({}))
        sol = M.solve_dc(bld, {}, M.MNASpec(mode="dcop", temp=27.0))
        vds = float(sol["out"])
        ids = 1e-3 / 2 * 0.5 ** 2 * (1 + 0.02 * vds)
        assert vds > 0.5 and abs((5.0 - vds) / 10e3 - ids) < 1e-12              # saturation, KCL at the drain
        gm, gds = 1e-3 * 0.5 * (1 + 0.02 * vds), 1e-3 / 2 * 0.5 ** 2 * 0.02
        rout = 1.0 / (1.0 / 10e3 + gds)
        expected = (4 * KT / 10e3 + 4 * KT * (2.0 / 3.0) * gm + kf * ids ** af / freqs ** ffe) * rout ** 2
        on, c, _, _ = M.noise(bld, {}, M.MNASpec(temp=27.0), "out", freqs)
        assert np.allclose(on, expected, rtol=1e-6)
        assert np.allclose(c["rd"], 4 * KT / 10e3 * rout ** 2, rtol=1e-6) and np.allclose(c["m1"], expected - c["rd"], rtol=1e-6)
        ns = product_noise_at_the_oracle_point(circ, "out", freqs)
        assert np.allclose(ns["onoise"], on, rtol=1e-12) and np.allclose(ns["m1"], c["m1"], rtol=1e-12)
    cut = _common_source()
    cut.devices[1].params["dc"] = 0.2                                           # cutoff: the channel injects nothing
    on, c, _, _ = M.noise(make_builder(cut.to_dicts({})), {}, M.MNASpec(temp=27.0), "out", freqs)
    assert "m1" not in c and np.allclose(on, 4 * KT * 10e3, rtol=1e-6)


@pytest.mark.parametrize("kind", ["D", "DCAP"])
def test_builtin_diode_flicker_noise(kind):
    """Diode / DiodeWithCap with KF > 0: shot noise 2q|I0| plus KF |I0|^AF / f^FFE at the junction bias (devices.jl:1393-1443, 1582-1585);
    both under the device's name.  A forward-biased diode behind 10 kOhm: S_out = (4kT / R + 2q I0 + KF I0^AF / f^FFE) (R || rd)^2."""
    from cadnip_jl_amd.circuit import Circuit
    c = Circuit("diode flicker")
    c.V("v1", "in", "0", dc=5.0)
    c.R("r1", "in", "out", 10e3)
    if kind == "D":
        c.D("d1", "out", "0", Is=1e-14, KF=1e-15, AF=1.5, FFE=1.1)
    else:
        c.DCAP("d1", "out", "0", Is=1e-14, Cj0=0.0, KF=1e-15, AF=1.5, FFE=1.1)
    freqs = np.array([1.0, 100.0, 1e4])
    bld = make_builder(c.to_dicts({}))
    sol = M.solve_dc(bld, {}, M.MNASpec(mode="dcop", temp=27.0))
    vd = float(sol["out"])
    i0 = 1e-14 * (np.exp(vd / 0.026) - 1.0)
    rout = 1.0 / (1.0 / 10e3 + 1e-14 / 0.026 * np.exp(vd / 0.026))
    expected = (4 * KT / 10e3 + 2 * M.Q_ELEMENTARY * i0 + 1e-15 * i0 ** 1.5 / freqs ** 1.1) * rout ** 2
    on, cc, _, _ = M.noise(bld, {}, M.MNASpec(temp=27.0), "out", freqs)
    assert np.allclose(on, expected, rtol=1e-5) and on[0] > 1.5 * on[2]          # the flicker term is visible at 1 Hz
    ns = product_noise_at_the_oracle_point(c, "out", freqs)
    assert np.allclose(ns["onoise"], on, rtol=1e-10) and np.allclose(ns["d1"], cc["d1"], rtol=1e-10)
