"""CPU suite, part 1: the oracle pinned against the reference's own fixtures.

Every test cites the reference test it restates (paths relative to /root/reference).  These are the
closed-form / exact-entry known answers of SURVEY.md section 8c; they are data (numbers), not source.
"""
import math

import numpy as np
import pytest

from oracle import mna_ref as M
from oracle import devices_ref as D
from oracle.netlist_ref import make_builder
from tests import circuits as tc


def dense(sys_):
    return sys_.G.toarray(), sys_.C.toarray()


# ---- test/mna/core.jl:197-234 stamping primitives -------------------------------------------------------
def test_stamping_primitives():
    ctx = M.MNAContext()
    n1, n2 = ctx.get_node("n1"), ctx.get_node("n2")
    ctx.stamp_G(n1, n1, 1.0); ctx.stamp_G(n1, n2, -1.0); ctx.stamp_G(n2, n1, -1.0); ctx.stamp_G(n2, n2, 1.0)
    assert ctx.G_I == [1, 1, 2, 2] and ctx.G_J == [1, 2, 1, 2] and ctx.G_V == [1.0, -1.0, -1.0, 1.0]
    ctx.stamp_C(n1, n1, 1e-6)
    assert len(ctx.C_V) == 1
    ctx.stamp_b(n1, 5.0)
    assert M.get_rhs(ctx)[0] == 5.0
    ctx.stamp_b(n1, 3.0)
    assert M.get_rhs(ctx)[0] == 8.0
    ctx.stamp_G(0, n1, 1.0); ctx.stamp_G(n1, 0, 1.0); ctx.stamp_b(0, 5.0)   # ground ignored
    assert len(ctx.G_V) == 4


# ---- test/mna/core.jl:239-505 device stamp patterns --------------------------------------------------------
def test_resistor_capacitor_patterns():
    ctx = M.MNAContext()
    n1, n2 = ctx.get_node("n1"), ctx.get_node("n2")
    D.stamp_resistor(ctx, n1, n2, 1000.0)
    G, C = dense(M.assemble(ctx))
    assert np.allclose(G, [[1e-3, -1e-3], [-1e-3, 1e-3]]) and M.assemble(ctx).G.nnz == 4
    ctx = M.MNAContext()
    n1, n2 = ctx.get_node("n1"), ctx.get_node("n2")
    D.stamp_capacitor(ctx, n1, n2, 1e-6)
    s = M.assemble(ctx)
    assert s.G.nnz == 0 and np.allclose(s.C.toarray(), [[1e-6, -1e-6], [-1e-6, 1e-6]])


def test_vsource_isource_inductor_patterns():
    ctx = M.MNAContext()
    vcc = ctx.get_node("vcc")
    I = D.stamp_vsource(ctx, vcc, 0, 5.0, name="V1")
    assert ctx.resolve_index(I) == 2
    s = M.assemble(ctx)
    G = s.G.toarray()
    assert G[0, 1] == 1.0 and G[1, 0] == 1.0 and s.b[1] == 5.0
    ctx = M.MNAContext()
    n1 = ctx.get_node("n1")
    D.stamp_isource(ctx, n1, 0, 0.001)
    s = M.assemble(ctx)
    assert s.G.nnz == 0 and s.b[0] == pytest.approx(0.001)
    ctx = M.MNAContext()
    n1, n2 = ctx.get_node("n1"), ctx.get_node("n2")
    I = D.stamp_inductor(ctx, n1, n2, 1e-3, "L1")
    assert ctx.resolve_index(I) == 3
    G, C = dense(M.assemble(ctx))
    assert (G[0, 2], G[1, 2], G[2, 0], G[2, 1]) == (1.0, -1.0, 1.0, -1.0) and C[2, 2] == pytest.approx(-1e-3)


def test_controlled_source_patterns():
    def four():
        ctx = M.MNAContext()
        return ctx, [ctx.get_node(n) for n in ("out_p", "out_n", "in_p", "in_n")]
    ctx, nd = four()
    D.stamp_vccs(ctx, *nd, 0.01)
    G = M.assemble(ctx).G.toarray()
    assert (G[0, 2], G[0, 3], G[1, 2], G[1, 3]) == (-0.01, 0.01, 0.01, -0.01)
    ctx, nd = four()
    I = D.stamp_vcvs(ctx, *nd, 10.0, "E1")
    assert ctx.resolve_index(I) == 5
    G = M.assemble(ctx).G.toarray()
    assert (G[0, 4], G[1, 4], G[4, 0], G[4, 1], G[4, 2], G[4, 3]) == (1.0, -1.0, 1.0, -1.0, -10.0, 10.0)
    ctx, nd = four()
    Iout, Iin = D.stamp_ccvs(ctx, *nd, 1000.0, "H1")
    assert ctx.resolve_index(Iin) == 5 and ctx.resolve_index(Iout) == 6
    G = M.assemble(ctx).G.toarray()
    assert (G[2, 4], G[3, 4], G[4, 2], G[4, 3], G[0, 5], G[1, 5], G[5, 0], G[5, 1], G[5, 4]) == (1, -1, 1, -1, 1, -1, 1, -1, -1000.0)
    ctx, nd = four()
    Iin = D.stamp_cccs(ctx, *nd, 2.0, "F1")
    assert ctx.resolve_index(Iin) == 5
    G = M.assemble(ctx).G.toarray()
    assert (G[2, 4], G[3, 4], G[4, 2], G[4, 3], G[0, 4], G[1, 4]) == (1, -1, 1, -1, -2.0, 2.0)


# ---- test/mna/precompile.jl:18-68 COO -> nz mapping ----------------------------------------------------------
def test_coo_to_csc_mapping_with_duplicates():
    import scipy.sparse as sp
    I, J, V = [1, 2, 1, 3, 2], [1, 1, 2, 2, 3], [1.0, 2.0, 3.0, 4.0, 5.0]
    S = sp.coo_matrix((V, (np.array(I) - 1, np.array(J) - 1)), shape=(3, 3)).tocsc()
    m = M.compute_coo_to_nz_mapping(I, J, S)
    nz = np.zeros(S.nnz)
    for k, v in enumerate(V):
        nz[m[k] - 1] += v
    assert np.allclose(nz, S.data)
    I, J, V = [1, 1, 2], [1, 1, 2], [1.0, 2.0, 3.0]
    S = sp.coo_matrix((V, (np.array(I) - 1, np.array(J) - 1)), shape=(2, 2)).tocsc()
    S.sum_duplicates()
    m = M.compute_coo_to_nz_mapping(I, J, S)
    assert m[0] == m[1] and m[2] > 0


# ---- README.md:50-57, test/mna/precompile.jl:109-137, 244-275 ------------------------------------------------------
def test_divider_and_source_current():
    sol = M.dc(make_builder(tc.divider().to_dicts()))
    assert sol.converged and sol["out"] == pytest.approx(2.5, abs=1e-10)
    assert sol["I_v1"] == pytest.approx(-2.5e-3, abs=1e-12)
    sol = M.dc(make_builder(tc.divider(10.0).to_dicts()))
    assert sol["out"] == pytest.approx(5.0, abs=1e-10)
    for v, r, i in ((10.0, 1000.0, -0.01), (20.0, 1000.0, -0.02)):
        devs = [{"type": "V", "name": "v", "nodes": ["a", "0"], "dc": v}, {"type": "R", "name": "r", "nodes": ["a", "0"], "r": r}]
        assert M.dc(make_builder(devs))["I_v"] == pytest.approx(i, abs=1e-10)


# ---- test/mna/precompile.jl:169-242 --------------------------------------------------------------------------------
def test_residual_zero_at_dc_point_and_structure_invariance():
    b = make_builder(tc.diode_rectifier().to_dicts())
    spec = M.MNASpec(mode="dcop")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    assert cs.n == 4   # 2 nodes + 1 current + 1 limit variable
    u, ok, it = M.dc_pcnr_newton(cs, ws, np.zeros(4))
    assert ok
    r = np.zeros(4)
    M.fast_residual(r, np.zeros(4), u, ws, 0.0)
    assert np.linalg.norm(r) < 1e-10
    nnz0 = cs.G.nnz
    for x in (np.zeros(4), np.array([5.0, 0.7, -1e-3, 0.7]), np.array([1.0, -2.0, 0.0, -2.0])):
        M.fast_rebuild(ws, x, 0.0)
        assert cs.G.nnz == nnz0 and not ws.dctx.overflow


# ---- test/mna/pcnr.jl:114-172 -------------------------------------------------------------------------------------------
def test_pnjlim_known_answers():
    assert D.pnjlim(0.5, 0.49, 0.026, 0.7) == (0.5, False)
    for v in (-0.3, 0.0, 0.3, 0.6588):
        vl, lim = D.pnjlim(v, v, 0.026, 0.6588)
        assert vl == pytest.approx(v) and lim is False
    vl, lim = D.pnjlim(5.0, 0.6, 0.026, 0.6588)
    assert lim and 0.6 < vl < 5.0 and vl < 1.0
    vl, lim = D.pnjlim(5.0, 0.0, 0.026, 0.66)
    assert lim and vl == pytest.approx(0.026 * math.log(5.0 / 0.026))
    assert D.pnjlim(-10.0, 0.5, 0.026, 0.66) == (-1.5, True)
    assert D.pnjlim(-0.5, 0.0, 0.026, 0.66) == (-0.5, False)


def test_diode_iv_linear_extension():
    Is, nVt = 1e-14, 0.026
    Ilo, Glo = D.diode_iv(Is, nVt, 80.0 * nVt - 1e-9)
    Ihi, Ghi = D.diode_iv(Is, nVt, 80.0 * nVt + 1e-9)
    assert Ilo == pytest.approx(Ihi, rel=1e-6) and Glo == pytest.approx(Ghi, rel=1e-6)
    I1, G1 = D.diode_iv(Is, nVt, 10.0)
    I2, G2 = D.diode_iv(Is, nVt, 11.0)
    assert math.isfinite(I1) and G1 == G2 and (I2 - I1) == pytest.approx(G1, rel=1e-12)
    I3, G3 = D.diode_iv(Is, nVt, 0.7)
    assert I3 == pytest.approx(Is * (math.exp(0.7 / nVt) - 1.0)) and G3 == pytest.approx(Is / nVt * math.exp(0.7 / nVt))


# ---- test/mna/pcnr.jl:330-387, doc/pcnr_plan.md:469-470 --------------------------------------------------------------
@pytest.mark.parametrize("mk", [tc.diode_rectifier, tc.diode_chain])
def test_pcnr_converges_within_ten_iterations(mk):
    b = make_builder(mk().to_dicts())
    spec = M.MNASpec(mode="dcop")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    u, ok, it = M.dc_pcnr_newton(cs, ws, np.zeros(cs.n), abstol=1e-10)
    assert ok and it <= 10
    if mk is tc.diode_rectifier:
        assert it == 7           # the reference measured 7
    u2, ok2, it2 = M.dc_pcnr_newton(cs, ws, 0.99 * u, abstol=1e-10)   # warm start converges in fewer
    assert ok2 and it2 <= it
    # fixed-point invariance: limit=true and limit=false agree on the node voltages
    bb = make_builder(tc.diode_rectifier(False).to_dicts())
    assert M.dc(bb)["out"] == pytest.approx(M.dc(make_builder(tc.diode_rectifier().to_dicts()))["out"], abs=1e-9)


# ---- test/transients.jl:112-131, 301-393 ; src/mna/devices.jl:47-103 -------------------------------------------------
def test_pulse_repeats_and_pwl():
    v1, v2, td, tr, tf, pw, per = 0.0, 1.0, 1e-3, 1e-6, 1e-6, 2e-3, 5e-3
    hi = tr + pw / 2
    lo = tr + tf + pw + (per - (tr + tf + pw)) / 2
    P = lambda t: D.pulse_at_time(v1, v2, td, tr, tf, pw, per, t)
    assert P(0.0) == v1 and P(td + hi) == v2 and P(td + per + hi) == v2 and P(td + 3 * per + hi) == v2 and P(td + 2 * per + lo) == v1
    ts, ys = [0.0, 1.0, 2.0, 2.0, 3.0], [0.0, 1.0, 1.0, 3.0, 3.0]
    assert D.pwl_at_time(ts, ys, -1.0) == 0.0 and D.pwl_at_time(ts, ys, 0.5) == 0.5 and D.pwl_at_time(ts, ys, 1.5) == 1.0
    assert D.pwl_at_time(ts, ys, 2.5) == 3.0 and D.pwl_at_time(ts, ys, 9.0) == 3.0


def test_breakpoint_expansion():
    # test/transients.jl:325-331: PulseWave(0, 1, td=1us, 1us, 1us, 3us, per=10us) over (0, 30us) -> 12 edges
    pulse = D.PulseWave(0.0, 1.0, 1e-6, 1e-6, 1e-6, 3e-6, 10e-6).breakpoints()
    bps = D.expand_breakpoints([pulse], (0.0, 30e-6))
    assert len(bps) == 12
    assert bps == sorted(bps) and all(0.0 < t < 30e-6 for t in bps)
    pwl = D.PWLWave([0.0, 1e-3, 2e-3], [0, 1, 0]).breakpoints()
    assert D.expand_breakpoints([pwl, pwl], (0.0, 5e-3)) == [1e-3, 2e-3]     # duplicates collapse, endpoints excluded
    from cadnip_jl_amd.structure import expand_breakpoints, wave_breakpoints
    assert expand_breakpoints([wave_breakpoints(("pulse", 0.0, 1.0, 1e-6, 1e-6, 1e-6, 3e-6, 10e-6))], (0.0, 30e-6)) == bps


# ---- test/sweep.jl:299-312 -----------------------------------------------------------------------------------------------
def test_dc_sweep_grid_on_oracle():
    for r1 in np.linspace(100, 2000, 4):
        for r2 in np.linspace(100, 2000, 4):
            devs = [{"type": "V", "name": "v", "nodes": ["vcc", "0"], "dc": 1.0},
                    {"type": "R", "name": "r1", "nodes": ["vcc", "out"], "r": "r1"},
                    {"type": "R", "name": "r2", "nodes": ["out", "0"], "r": "r2"}]
            sol = M.solve_dc(make_builder(devs), {"r1": r1, "r2": r2}, M.MNASpec(mode="dcop"))
            assert sol["I_v"] == pytest.approx(-1.0 / (r1 + r2), abs=1e-7)


# ---- test/mna/va_mosfet.jl:329-461 (logic levels of a CMOS inverter; there: square-law VA, atol 0.01) -------------------------
def test_cmos_inverter_logic_levels():
    for vin, lo, hi in ((0.0, 4.99, 5.01), (5.0, -0.01, 0.01)):
        sol = M.dc(make_builder(tc.inverter_dc(vin).to_dicts()))
        assert sol.converged and lo < sol["out"] < hi


# ---- sp_mos1 numerics: the reference's own numeric fixtures for the level-1 model ---------------------------------------
def _dc_system(circ):
    """dc! on the oracle, then the linearisation at the solution: dense G, C and the MNAData (names)."""
    b = make_builder(circ.to_dicts())
    sol = M.dc(b)
    assert sol.converged
    spec = M.MNASpec(mode="dcop")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    M.fast_rebuild(ws, sol.x, 0.0)
    return sol, cs.G.toarray(), cs.C.toarray()


# test/opinfo.jl:158-205 (hand derivation :158-163): K = kp W/L = 2 mA/V^2, VOV = 447.2 mV, ID = 200 uA, gm = K VOV = 894.4 uS
def test_mos1_cs_stage_small_signal_numbers():
    VOV = 1.1472 - 0.7
    GM = 100e-6 * 20.0 * VOV
    sol, G, _ = _dc_system(tc.cs_stage())
    ix = lambda nm: sol.sys.index_of(nm) - 1
    d, g = ix("drain"), ix("gate")
    i_d = -sol["I_vdd"]                                       # opinfo.jl:92: i_m1_d == -I_vdd
    assert i_d == pytest.approx(200e-6, rel=0.05)             # opinfo.jl:89
    assert i_d == pytest.approx((5.0 - sol["drain"]) / 10e3, rel=1e-6)   # opinfo.jl:94 (i_rd_p)
    gm = G[d, g]                                              # dI_d/dV_g: the drain row's gate column carries nothing else
    assert gm == pytest.approx(GM, rel=0.05)                  # opinfo.jl:183
    up = -M.dc(make_builder(tc.cs_stage(1.1472 + 1e-3).to_dicts()))["I_vdd"]
    dn = -M.dc(make_builder(tc.cs_stage(1.1472 - 1e-3).to_dicts()))["I_vdd"]
    assert gm == pytest.approx((up - dn) / 2e-3, rel=0.02)    # opinfo.jl:187-190: gm is the derivative of the drain current
    gds = G[d, d] - 1.0 / 10e3                                # drain diagonal minus Rd (bulk junction: gmin-sized)
    assert gds == pytest.approx(0.01 * i_d, rel=0.10)         # opinfo.jl:193
    # vgs / vds as the model sees them are the settled $limit variables (opinfo.jl:199-200)
    assert sol["m1_sp_mos1_lim_g_s_int"] == pytest.approx(1.1472, rel=1e-6)
    assert sol["m1_sp_mos1_lim_d_int_s_int"] == pytest.approx(sol["drain"], rel=1e-6)
    assert sol["drain"] > VOV                                 # saturated: vds clears vdsat (opinfo.jl:198)
    hot = -M.dc(make_builder(tc.cs_stage(1.20).to_dicts()))["I_vdd"]
    assert hot > i_d * 1.1                                    # opinfo.jl:142-145


def load_ngspice_inverter():
    import os
    rows = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "ngspice43_cmos_inverter_ac.csv"), delimiter=",", comments="#")
    return rows[:, 0], rows[:, 1] + 1j * rows[:, 2]


def check_against_ngspice(resp, ref):
    assert np.allclose(np.abs(resp), np.abs(ref), rtol=0.05, atol=0.0)           # test/ac.jl:267
    dph = np.abs(np.angle(resp) - np.angle(ref))
    assert np.all(np.minimum(dph, 2 * np.pi - dph) < 0.1)                        # test/ac.jl:270-272


# test/ac.jl:204-272: sp_mos1 CMOS inverter, H(jw) = (G + jw C)^-1 b_ac at the DC point vs the ngspice-43 table
def test_mos1_inverter_ac_matches_ngspice_table():
    freqs, ref = load_ngspice_inverter()
    sol, G, C = _dc_system(tc.cmos_inverter_ac())
    b_ac = np.zeros(G.shape[0])
    b_ac[sol.sys.index_of("I_vin") - 1] = 1.0                 # "AC 1" on Vin: unit excitation on its branch row (get_rhs_ac)
    resp = M.ac_response(G, C, b_ac, 2 * np.pi * freqs, sol.sys.n_nodes)[:, sol.sys.index_of("vout") - 1]
    check_against_ngspice(resp, ref)
    # far inside the reference's own 5 % / 0.1 rad: the table's six digits are reproduced (measured 3.4e-6 / 9e-7 rad)
    assert np.allclose(np.abs(resp), np.abs(ref), rtol=1e-4, atol=0.0) and np.max(np.abs(np.angle(resp) - np.angle(ref))) < 1e-4    # the low-frequency gain -(gmn + gmp) / (gdsn + gdsp)


# ---- solve.jl:720-929: shape of the fallback ladders on the oracle (the GPU side is compared rung for rung in
# tests/test_gpu_dc_fallbacks.py) ----------------------------------------------------------------------------------------------
def test_dc_fallback_ladders_on_oracle():
    from tests.dc_chain_util import oracle_chain, same_ladder

    def chain(circ, **kw):
        b = make_builder(circ.to_dicts())
        spec = M.MNASpec(mode="dcop")
        ctx = M.build_with_detection(b, {}, spec)
        cs = M.compile_structure(b, {}, spec, ctx=ctx)
        return oracle_chain(cs, M.create_workspace(cs, ctx=ctx), np.zeros(cs.n), **kw)

    # no limiting: plain Newton spends its 100 iterations, the gshunt ladder 1e-3, /10 ... 1e-12 and the solve at the target arrive
    u, ok, log = chain(tc.diode_rectifier(False))
    assert ok and log[0] == (1, 0.0, False, 100)
    rungs = [v for s, v, _, _ in log[1:]]
    # (repeated division leaves 1.0000000000000002e-12 > 1e-12 on the tenth rung, so an eleventh at exactly 1e-12 follows)
    assert all(s == 2 and good for s, _, good, _ in log[1:]) and len(rungs) == 12
    assert np.allclose(rungs[:10], [10.0 ** -k for k in range(3, 13)], rtol=1e-12) and rungs[10] == 1e-12 and rungs[11] == 0.0
    assert u[1] == pytest.approx(M.dc(make_builder(tc.diode_rectifier().to_dicts()))["out"], abs=1e-9)
    # a failed rung is retried from the saved solution with sqrt(factor): 10 -> 3.16 -> 1.78 -> 1.33 <= 1.5 gives up after 4
    u, ok, log = chain(tc.bi_quadratic(), abstol=1e-7, maxiters=5)
    assert ok and [e[:3] for e in log[:5]] == [(1, 0.0, False)] + [(2, 1e-3, False)] * 4
    assert all(s == 3 and good for s, _, good, _ in log[5:]) and log[5][1] == 0.0 and log[-1][1] == 1.0
    assert same_ladder([(0,) + log[0][1:]] + log[1:], log) and not same_ladder(log[:-1], log)
