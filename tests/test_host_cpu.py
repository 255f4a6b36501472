"""CPU suite, part 2: product host logic, the C ABI's symbol table, the LU program, and the
oracle's two restatements against each other.  No compute call touches a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import cadnip_jl_amd as cj
from cadnip_jl_amd import api, benchmarks as bm, hip
from cadnip_jl_amd.structure import TYPE_ID
from oracle import mna_ref as M
from oracle.netlist_ref import make_builder
from tests import circuits as tc
from tests.port_util import make_port, analyze_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cadnip_hip.h")).read()
    declared = set(re.findall(r"\b(cadnip_[A-Za-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = hip.load_library()
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libcadnip_hip.so does not export %s" % sym
    assert declared == set(hip.EXPORTS)
    assert b"gfx950" in lib.cadnip_version()


@pytest.mark.parametrize("name", list(tc.ALL_STAMP))
def test_product_structure_matches_oracle_discovery(name):
    """Product structure discovery (device table) == oracle's literal builder passes: sizes, names, the
    unified pattern in the reference's CSC order, COO counts."""
    mk, params = tc.ALL_STAMP[name]
    circ = mk()
    st = cj.discover(circ, params)
    b = make_builder(circ.to_dicts(params))
    spec = M.MNASpec(mode="tran")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits) == (cs.n, ctx.n_nodes, ctx.n_currents, ctx.n_charges, ctx.n_limits)
    assert st.node_names == ctx.node_names and st.current_names == ctx.current_names
    assert st.charge_names == ctx.charge_names and st.limit_names == ctx.limit_names
    assert np.array_equal(st.ref_colptr, cs.colptr) and np.array_equal(st.ref_rowval, cs.rowval)
    assert (st.n_coo_g, st.n_coo_c, st.n_coo_b) == (cs.G_n_coo, cs.C_n_coo, cs.n_b_deferred)
    assert np.allclose(st.limit_init, cs.limit_init)
    # CSR <-> reference nz permutation is a bijection consistent with the pattern
    rows = np.repeat(np.arange(st.n), np.diff(st.rowptr))
    assert np.array_equal(st.ref_rowval[st.to_ref_nz], rows)
    assert sorted(st.to_ref_nz.tolist()) == list(range(st.nnz))


def test_dff_sizes():
    st = cj.discover(bm.dff_circuit(meyer=True), {"vdd": 5.0})
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits) == (265, 18, 7, 120, 120)   # SURVEY.md appendix A.2 estimate
    st = cj.discover(bm.dff_circuit(), {"vdd": 5.0})
    assert st.n == 18 + 7 + 90 + 120 and st.mos1_vdep == (False, True, True, True)


@pytest.mark.parametrize("name", ["linear_zoo", "nonlinear_zoo", "behavioral", "inverter", "mos1_rd", "dff", "dff_meyer"])
def test_cpu_port_stamps_match_literal_oracle(name):
    """The compiled port (Dual<3>, hoisted setup/temp) vs the literal Python restatement (width-14 duals,
    per-call setup/temp): G, C, b, limit_w at random operating points, several temperatures."""
    mk, params = tc.ALL_STAMP[name]
    circ = mk()
    rng = np.random.default_rng(5)
    for temp, t, mode in ((27.0, 0.0, "tran"), (-40.0, 1.3e-3 if "dff" not in name else 2.005e-7, "tran"), (125.0, 0.0, "dcop")):
        st, port = make_port(circ, params, temp, mode)
        b = make_builder(circ.to_dicts(params))
        spec = M.MNASpec(mode=mode, temp=temp)
        ctx = M.build_with_detection(b, {}, spec)
        cs = M.compile_structure(b, {}, spec, ctx=ctx)
        ws = M.create_workspace(cs, ctx=ctx)
        for trial in range(2):
            u = rng.random(st.n) * (5.0 if trial else 1.0) - 0.3
            M.fast_rebuild(ws, u, t)
            G, C, bb, lw = port.rebuild(u, t)
            Gr, Cr = np.empty(st.nnz), np.empty(st.nnz)
            Gr[st.to_ref_nz], Cr[st.to_ref_nz] = G, C
            for got, ref in ((Gr, cs.G.data), (Cr, cs.C.data), (bb, ws.dctx.b)):
                assert np.max(np.abs(got - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1e-300), (name, temp, trial)
            if st.n_limits:
                assert np.max(np.abs(lw - ws.dctx.limit_w)) <= 1e-12 * max(1.0, np.max(np.abs(lw)))
        port.close()


@pytest.mark.parametrize("name", ["diode", "diode_chain", "inverter", "mos1_rd"])
def test_cpu_port_dc_matches_literal_oracle(name):
    mk, params = tc.ALL_STAMP[name]
    circ = mk()
    st, port = make_port(circ, params, 27.0, "dcop")
    analyze_port(st, port, 5.0)
    u, ok, it = port.dc(abstol=1e-10)
    b = make_builder(circ.to_dicts(params))
    spec = M.MNASpec(mode="dcop")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    uo, oko, ito = M.dc_pcnr_newton(cs, ws, np.zeros(cs.n), abstol=1e-10)
    assert ok and oko and it == ito
    assert np.max(np.abs(u - uo) / np.maximum(np.abs(uo), 1.0)) < 1e-9
    port.close()


def test_behavioral_sources_fixed_point():
    """Behavioural sources are stamped as fixed sources at the current iterate (devices.jl:1079-1131: b only, no
    Jacobian), so Newton on them is the fixed-point iteration x <- 2 - 0.1 x^2; its limit is the closed form
    x* = (-1 + sqrt(1.8)) / 0.2.  Literal oracle (plain Newton loop, solve.jl:542-578) and the C++ port agree."""
    circ = tc.behavioral()
    b = make_builder(circ.to_dicts({}))
    spec = M.MNASpec(mode="dcop")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    uo, oko, ito = M.dc_newton_plain(cs, ws, np.zeros(cs.n), abstol=1e-10, maxiters=200)
    xs = (-1.0 + np.sqrt(1.8)) / 0.2
    assert oko and abs(uo[cs.node_names.index("x")] - xs) < 1e-9
    y = 0.5 * (2 * xs + np.tanh(2.0 - xs) - min(xs, 0.5) * np.exp(-2.0))
    assert abs(uo[cs.node_names.index("y")] - y) < 1e-9
    st, port = make_port(circ, {}, 27.0, "dcop")
    analyze_port(st, port, 5.0)
    u, ok, it = port.dc(abstol=1e-10, maxiters=200)
    assert ok and np.max(np.abs(u - uo)) < 1e-9
    port.close()
    # the expression compiler rejects what the stamp kernels cannot interpret
    for bad in ("V(x) if 1 else 2", "foo(V(x))", "x + 1", "V(a, b, c)"):
        with pytest.raises((ValueError, SyntaxError)):
            cj.bsource.compile_expr(bad)


def test_cpu_port_rc_charge_analytic():
    # test/mna/core.jl:785-912: V(t) = 5 (1 - exp(-t/tau)) at t in {tau/2, tau, 2tau, 3tau, 5tau}, rtol 1e-3 (DAE path)
    c = cj.Circuit()
    c.V("v1", "vin", "0", dc=0.0, wave=("pwl", [0.0, 1e-9], [0.0, 5.0]))
    c.R("r1", "vin", "out", 1e3)
    c.C("c1", "out", "0", 1e-6)
    st, port = make_port(c, {}, 27.0, "tranop")
    analyze_port(st, port, 5.0)
    u0, ok, _ = port.dc(abstol=1e-9)
    assert ok
    port.set_spec(mode="tran")
    tau = 1e-3
    ts = np.array([0.5, 1.0, 2.0, 3.0, 5.0]) * tau
    out, _, stats, _ = port.tran(u0, 0.0, 5e-3, 1e-9, 1e-6, breaks=[1e-9], save_t=ts, obs=[st.index_of("out")], err_mask=st.differential_mask())
    assert stats["status"] == 1
    assert np.allclose(out[:, 0], 5.0 * (1 - np.exp(-ts / tau)), rtol=1e-3)
    port.close()


def test_cpu_port_bdf3_accuracy_and_stability():
    """max_order = 3 in the port (the oracle side of CadnipTranOpts.max_order = 3, mirrored operation for operation in csrc/tran_ctrl.hpp): on the
    RC step response it needs fewer steps than BDF2 for a smaller error against 5 (1 - exp(-t / tau)) (test/mna/core.jl:785-912's closed form),
    and on a lightly damped series RLC (Q = 3.2) it stays stable and tracks the ringing -- BDF3 is not A-stable, the controller has to keep it honest."""
    c = cj.Circuit()
    c.V("v1", "vin", "0", dc=0.0, wave=("pwl", [0.0, 1e-9], [0.0, 5.0]))
    c.R("r1", "vin", "out", 1e3)
    c.C("c1", "out", "0", 1e-6)
    st, port = make_port(c, {}, 27.0, "tranop")
    analyze_port(st, port, 5.0)
    u0, ok, _ = port.dc(abstol=1e-9)
    port.set_spec(mode="tran")
    tau = 1e-3
    ts = np.array([0.5, 1.0, 2.0, 3.0, 5.0]) * tau
    res = {}
    for mo in (2, 3):
        out, _, stats, _ = port.tran(u0, 0.0, 5e-3, 1e-9, 1e-6, breaks=[1e-9], save_t=ts, obs=[st.index_of("out")], err_mask=st.differential_mask(), max_order=mo)
        assert stats["status"] == 1
        res[mo] = (stats["accepted"], np.max(np.abs(out[:, 0] - 5.0 * (1 - np.exp(-ts / tau))) / 5.0))
    port.close()
    assert res[3][0] < 0.6 * res[2][0] and res[3][1] < res[2][1] < 1e-4, res
    c = cj.Circuit()
    c.V("v1", "in", "0", dc=0.0, wave=("pwl", [0.0, 1e-9], [0.0, 1.0]))
    c.R("r1", "in", "a", 10.0)
    c.L("l1", "a", "out", 1e-3)
    c.C("c1", "out", "0", 1e-6)
    st, port = make_port(c, {}, 27.0, "tranop")
    analyze_port(st, port, 1.0)
    u0, ok, _ = port.dc(abstol=1e-9)
    port.set_spec(mode="tran")
    al, w0 = 10.0 / (2 * 1e-3), 1.0 / np.sqrt(1e-3 * 1e-6)
    wd = np.sqrt(w0 ** 2 - al ** 2)
    ts = np.linspace(1e-5, 2e-3, 40)
    exact = 1 - np.exp(-al * ts) * (np.cos(wd * ts) + al / wd * np.sin(wd * ts))
    out, _, stats, _ = port.tran(u0, 0.0, 2e-3, 1e-9, 1e-6, breaks=[1e-9], save_t=ts, obs=[st.index_of("out")], err_mask=st.differential_mask(), max_order=3)
    port.close()
    assert stats["status"] == 1 and np.max(np.abs(out[:, 0] - exact)) < 2e-4


def test_cpu_port_dff_transient_logic_pins():
    # test/gf180_dff.jl:29-33 logic-level pins (race-free ones; see tests/test_gpu_drivers.py)
    circ = bm.dff_circuit()
    st, port = make_port(circ, {"vdd": 5.0}, 27.0, "tranop")
    analyze_port(st, port, 5.0)
    u0, ok, _ = port.dc(abstol=1e-9)
    assert ok
    port.set_spec(mode="tran")
    from cadnip_jl_amd.structure import expand_breakpoints
    ts = np.array([t for t, _ in bm.DFF_Q_PINS])
    out, _, stats, _ = port.tran(u0, 0.0, 7e-7, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4,
                                 breaks=expand_breakpoints(st.breakpoints, bm.DFF_TSPAN), save_t=ts, obs=[st.index_of("Q")],
                                 err_mask=st.differential_mask())
    assert stats["status"] == 1
    q = out[:, 0]
    assert abs(q[0]) < 0.05 and abs(q[1]) < 0.05 and abs(q[4] - 5.0) < 0.05
    port.close()


def _emulate_lu(P, vals, rhs, n):
    lu = np.zeros(P["rowptr"][-1])
    lu[P["load_dst"]] = vals[P["load_src"]]
    for lev in range(len(P["lev_ptr"]) - 1):
        new = {}
        for e in range(P["lev_ptr"][lev], P["lev_ptr"][lev + 1]):
            acc = lu[P["ent_pos"][e]]
            for t in range(P["ent_ptr"][e], P["ent_ptr"][e + 1]):
                acc -= lu[P["term_a"][t]] * lu[P["term_b"][t]]
            if P["ent_diag"][e] >= 0:
                acc /= lu[P["ent_diag"][e]]
            new[P["ent_pos"][e]] = acc
        for k, v in new.items():       # level-synchronous, like the GPU kernel
            lu[k] = v
    y = rhs[P["rperm"]].copy()
    for lev in range(len(P["fwd_lev_ptr"]) - 1):
        new = {}
        for r in range(P["fwd_lev_ptr"][lev], P["fwd_lev_ptr"][lev + 1]):
            i = P["fwd_rows"][r]
            new[i] = y[i] - sum(lu[p] * y[P["col"][p]] for p in range(P["rowptr"][i], P["diag"][i]))
        for k, v in new.items():
            y[k] = v
    for lev in range(len(P["bwd_lev_ptr"]) - 1):
        new = {}
        for r in range(P["bwd_lev_ptr"][lev], P["bwd_lev_ptr"][lev + 1]):
            i = P["bwd_rows"][r]
            new[i] = (y[i] - sum(lu[p] * y[P["col"][p]] for p in range(P["diag"][i] + 1, P["rowptr"][i + 1]))) / lu[P["diag"][i]]
        for k, v in new.items():
            y[k] = v
    x = np.zeros(n)
    x[P["cperm"]] = y
    return x


def test_symbolic_lu_program_solves_mna_systems():
    """Host symbolic phase (Markowitz order, fill, level schedule): emulate the GPU kernel's level-synchronous
    execution in numpy and check backward error; zero diagonals (V-source rows) included."""
    rng = np.random.default_rng(1)
    for n in (5, 40, 150):
        A = sp.random(n, n, density=min(0.5, 4.0 / n), random_state=int(rng.integers(1 << 30)), format="lil")
        for i in range(n):
            if i % 3:
                A[i, i] = rng.random() + 1
            A[i, (i + 1) % n] = rng.random() - 0.5
            A[(i + 1) % n, i] = rng.random() - 0.5
        A = A.tocsr()
        A.sort_indices()
        P = hip.host_lu_analyze(n, A.indptr, A.indices, A.data)
        rhs = rng.random(n)
        x = _emulate_lu(P, A.data, rhs, n)
        assert np.max(np.abs(A @ x - rhs) / (abs(A) @ np.abs(x) + 1.0)) < 1e-9   # threshold (1e-3) pivoting allows growth
        # every LU entry is written by exactly one program entry, in an order consistent with its inputs
        pos_level = {}
        for lev in range(len(P["lev_ptr"]) - 1):
            for e in range(P["lev_ptr"][lev], P["lev_ptr"][lev + 1]):
                assert P["ent_pos"][e] not in pos_level
                pos_level[P["ent_pos"][e]] = lev
        for lev in range(len(P["lev_ptr"]) - 1):
            for e in range(P["lev_ptr"][lev], P["lev_ptr"][lev + 1]):
                deps = list(P["term_a"][P["ent_ptr"][e]:P["ent_ptr"][e + 1]]) + list(P["term_b"][P["ent_ptr"][e]:P["ent_ptr"][e + 1]])
                if P["ent_diag"][e] >= 0:
                    deps.append(P["ent_diag"][e])
                assert all(pos_level.get(d, -1) < lev for d in deps)


def test_symbolic_lu_dff_has_almost_no_fill():
    st, port = make_port(bm.dff_circuit(), {"vdd": 5.0})
    prog = analyze_port(st, port, 5.0)
    assert prog["rowptr"][-1] <= st.nnz + 40
    port.close()


def test_singular_matrix_is_reported():
    A = sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 4.0]]))
    with pytest.raises(hip.SingularException):
        hip.host_lu_analyze(2, A.indptr, A.indices, A.data)


def test_sweep_algebra():
    # src/sweeps.jl:150-330 / test/sweep.jl: product order (first axis fastest), tandem, serial
    ps = api.ProductSweep(a=[1, 2, 3], b=[10, 20])
    assert list(ps) == [{"a": 1, "b": 10}, {"a": 2, "b": 10}, {"a": 3, "b": 10}, {"a": 1, "b": 20}, {"a": 2, "b": 20}, {"a": 3, "b": 20}]
    assert len(ps) == 6
    assert list(api.TandemSweep(a=[1, 2], b=[3, 4])) == [{"a": 1, "b": 3}, {"a": 2, "b": 4}]
    assert len(api.SerialSweep(api.Sweep(a=[1, 2]), api.Sweep(a=[5]))) == 3
    with pytest.raises(ValueError):
        api.TandemSweep(a=[1, 2], b=[3])
    mc = api.MNACircuit(tc.divider(), {})
    with pytest.raises(KeyError):
        api.CircuitSweep(mc, api.Sweep(nope=[1.0]))
    mc2 = api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0})
    assert api.alter(mc2, vdd=4.5).params["vdd"] == 4.5 and mc2.params["vdd"] == 5.0
    with pytest.raises(KeyError):
        api.alter(mc2, nope=1.0)


def test_pack_params_temperature_and_sweep_axes():
    circ = bm.dff_circuit()
    st = cj.discover(circ, {"vdd": 5.0})
    B = 3
    packed = cj.pack_params(st, circ, {"vdd": np.array([4.5, 5.0, 5.5])}, np.array([-40.0, 27.0, 125.0]), B)
    blk = {b.type: (b, p) for b, p in zip(st.blocks, packed)}
    vb, vp = blk["V"]
    assert vp.shape == (B, 2, vb.count)
    assert np.allclose(vp[:, 0, 0], [4.5, 5.0, 5.5])                   # VVDD dc follows vdd
    assert np.allclose(vp[:, 1, 5], np.array([4.5, 5.0, 5.5]) / 5.0)   # clock amplitude scale
    mb, mp = blk["MOS1"]
    from cadnip_jl_amd import mos1_params as m1
    kT = 1.38064852e-23 / 1.6021766208e-19
    assert np.allclose(mp[:, m1.P_VT, 0], (np.array([-40.0, 27.0, 125.0]) + 273.15) * kT)
    assert np.all(np.diff(mp[:, m1.P_BETA, 0]) < 0)                    # mobility falls with temperature


def test_spice_deck_reader_reproduces_the_dff_table():
    """The reference's own DFF deck (test/DFF/DFF_cap_all.cir + the gf180 cell netlist it includes, kept verbatim as data
    fixtures under tests/golden/) read by cadnip.jl_amd/netlist.py gives the hand-transcribed benchmark table: same
    instances, nodes, sizes, stimulus, voltage sources first (src/spc/codegen.jl:3130-3149)."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    deck = open(os.path.join(gold, "DFF_cap_all.cir")).read()
    cell = open(os.path.join(gold, "gf180mcu_fd_sc_mcu7t5v0__dffnq_4.ngspice")).read()
    circ, info = cj.netlist.read_spice(deck, models={"nfet_06v0": bm.NFET_06V0_MEYER, "pfet_06v0": bm.PFET_06V0_MEYER},
                                       includes={"gf180mcu_fd_sc_mcu7t5v0__dffnq_4.ngspice": cell})
    assert info["options"] == {"gmin": 1e-15} and info["tran"] == (3.3333333333333e-10, 6.0e-7)
    got, ref = circ.to_dicts({}), bm.dff_circuit(meyer=True).to_dicts({"vdd": 5.0})
    assert [d["name"] for d in got] == [d["name"] for d in ref]
    for g, r in zip(got, ref):
        assert g["type"] == r["type"] and g["nodes"] == r["nodes"], (g["name"], g, r)
        if g["type"] == "MOS1":
            assert g["model"] == pytest.approx(r["model"]) and g["m"] == r["m"]
        elif g["type"] == "C":
            assert g["c"] == pytest.approx(r["c"])
        else:
            assert g["dc"] == pytest.approx(r["dc"] * 1.0), g["name"]
            if "wave" in r:
                assert g["wave"][0] == "pwl" and np.allclose(g["wave"][1], r["wave"][1], rtol=1e-15) and np.allclose(g["wave"][2], np.array(r["wave"][2]) * r["scale"])
    # same structure as the benchmark table
    st_a, st_b = cj.discover(circ, {}), cj.discover(bm.dff_circuit(meyer=True), {"vdd": 5.0})
    assert st_a.n == st_b.n and st_a.nnz == st_b.nnz and st_a.node_names == st_b.node_names
    assert np.array_equal(st_a.colidx, st_b.colidx) and np.array_equal(st_a.g_slots, st_b.g_slots)


def test_spice_deck_reader_elements_and_errors():
    deck = """* rc + sources
    .param rload=2k
    V1 in 0 DC 1.5 PULSE(0 3.3 1n 1n 1n 5n 20n)   ; trailing comment
    I1 0 mid 1m
    R1 in mid {rload}
    R2 mid 0 cut
    C1 mid 0 10p
    L1 mid out 1u
    E1 e 0 mid 0 2.0
    G1 0 e in mid 1meg
    B1 out 0 I=V(mid)*1e-3
    B2 bv 0 V = 2*V(in)
    .tran 1n 100n
    .end
    R99 ignored 0 1
    """
    c, info = cj.netlist.read_spice(deck, sweep=("cut",))
    d = {x["name"]: x for x in c.to_dicts({"cut": 750.0})}
    assert list(d)[0] == "V1" and d["V1"]["dc"] == 1.5 and d["V1"]["wave"] == ("pulse", 0.0, 3.3, 1e-9, 1e-9, 1e-9, 5e-9, 2e-8)
    assert d["R1"]["r"] == 2000.0 and d["R2"]["r"] == 750.0 and d["C1"]["c"] == pytest.approx(1e-11) and d["G1"]["gm"] == 1e6
    assert d["B1"]["nodes"] == ["0", "out"] and d["B1"]["expr"] == "V(mid)*1e-3" and d["B2"]["type"] == "BV"
    assert "R99" not in d and info["tran"] == (1e-9, 1e-7)
    kick, _ = cj.netlist.read_spice("i0 0 1 dc 0 pulse 0 10u 1n 1n 1n 1n\nr1 1 0 1k\n")       # waveform without parentheses
    assert kick.to_dicts({})[0]["wave"] == ("pulse", 0.0, 1e-5, 1e-9, 1e-9, 1e-9, 1e-9, 0.0) and kick.to_dicts({})[0]["dc"] == 0.0
    assert cj.netlist.parse_number("2.5MEG") == 2.5e6 and cj.netlist.parse_number("10pF") == pytest.approx(1e-11)
    # .model cards: the ring oscillator deck of test/mna/vadistiller_integration.jl:45-60 in SPICE form == the table form
    ring = """* 3-stage ring
    .model pmos1 pmos level=1 vto=-0.7 kp=50e-6
    .model nmos1 nmos level=1 vto=0.7 kp=100e-6
    Vdd vdd 0 DC 3.3
    MP1 out1 in1 vdd vdd pmos1 w=2e-6 l=1e-6
    MN1 out1 in1 0 0 nmos1 w=1e-6 l=1e-6
    MP2 out2 out1 vdd vdd pmos1 w=2e-6 l=1e-6
    MN2 out2 out1 0 0 nmos1 w=1e-6 l=1e-6
    MP3 in1 out2 vdd vdd pmos1 w=2e-6 l=1e-6
    MN3 in1 out2 0 0 nmos1 w=1e-6 l=1e-6
    C1 out1 0 10f
    C2 out2 0 10f
    C3 in1 0 10f
    .END"""
    rc, _ = cj.netlist.read_spice(ring)
    ra, rb = rc.to_dicts({}), tc.ring_oscillator().to_dicts({})
    assert [(x["type"], x["name"].lower(), x["nodes"]) for x in ra] == [(x["type"], x["name"].lower(), x["nodes"]) for x in rb]
    assert all(x["model"] == pytest.approx(y["model"]) for x, y in zip(ra, rb) if x["type"] == "MOS1")
    with pytest.raises(ValueError):
        cj.netlist.read_spice(".model m1 nmos level=14")
    for bad in ("Q1 a b c npn", ".subckt x a b", "M1 d g s nfet"):
        with pytest.raises((ValueError, KeyError)):
            cj.netlist.read_spice(bad, models={})
    with pytest.raises(FileNotFoundError):
        cj.netlist.read_spice('.include "missing.sp"')


def _emulate_f2(P, n, csr_vals, rowptr, colidx, rhs):
    """Execute the fused kernel's linear-solve program (csrc/f2_program.cpp) the way k_fused2 does: work array W =
    [sparse L\\U | dense core | rhs | ...], passes of <= 64 lane descriptors (every read of a pass happens before any of its
    writes, exactly one writer per word and pass), the in-register dense solve of the core between the two pass lists."""
    f2 = P["f2"]
    nc, lu_words, dn0, n_pre, n_post = [int(v) for v in f2["meta"]]
    y0 = lu_words
    W = np.zeros(lu_words + n + 64)
    pinv = np.empty(n, dtype=int); pinv[P["rperm"]] = np.arange(n)
    for k in range(len(P["load_src"])):
        W[f2["posW"][P["load_dst"][k]]] += csr_vals[P["load_src"][k]]
    W[y0 + pinv] = rhs

    def run(first, count):
        for pi in range(first, first + count):
            pd = int(f2["passes"][pi])
            base, hi = pd & 0xFFFFFFFF, pd >> 32
            T, hasdiv, fence = hi & 0x7F, (hi >> 11) & 1, (hi >> 12) & 1
            writes = {}
            partial = {}
            for l in range(T):
                D = int(f2["lanes"][base + l])
                pos, dg, t0, dhi = D & 0xFFFF, (D >> 16) & 0xFFFF, (D >> 32) & 0xFFFF, D >> 48
                nt, lg, leader = dhi & 0xFF, (dhi >> 8) & 7, (dhi >> 12) & 1
                part = 0.0
                for t in range(nt):
                    tm = int(f2["terms"][t0 + t])
                    part += W[tm & 0xFFFF] * W[tm >> 16]
                grp = l >> lg                      # aligned group of 2^lg lanes
                key = (grp, lg)
                partial[key] = partial.get(key, 0.0) + part
                if leader:
                    assert l % (1 << lg) == 0
                    writes[key] = (pos, dg)
            for key, (pos, dg) in writes.items():
                acc = W[pos] - partial[key]
                if hasdiv and dg != 0xFFFF:
                    acc /= W[dg]
                assert pos not in [p for k2, (p, _) in writes.items() if k2 != key]
                writes[key] = (pos, acc)
            for key, (pos, acc) in writes.items():
                W[pos] = acc

    run(0, n_pre)
    if nc:
        S = W[dn0:dn0 + nc * nc].reshape(nc, nc).copy()
        b = W[y0 + n - nc:y0 + n].copy()
        for k in range(nc):                        # no pivoting: static order, as on the GPU
            for i in range(k + 1, nc):
                m = S[i, k] / S[k, k]
                S[i, k + 1:] -= m * S[k, k + 1:]
                b[i] -= m * b[k]
        for k in range(nc - 1, -1, -1):
            b[k] /= S[k, k]
            b[:k] -= S[:k, k] * b[k]
        W[y0 + n - nc:y0 + n] = b
    run(n_pre, n_post)
    x = np.empty(n)
    x[P["cperm"]] = W[y0:y0 + n]
    return x, (n_pre, n_post)


@pytest.mark.parametrize("nc", [0, 8, 12, 16])
def test_fused_linear_solve_program_on_cpu(nc):
    """The entry program the fused kernel executes (factorisation + forward substitution as an extra column, dense core,
    back substitution), emulated pass by pass on the DFF Jacobian and on random MNA-like matrices."""
    import scipy.sparse.linalg as spla
    st, port = make_port(bm.dff_circuit(), {"vdd": 5.0})
    rng = np.random.default_rng(11)
    u = rng.random(st.n) * 5.0
    G, Cm, b, lw = port.rebuild(u, 2.005e-7)
    J = G + 1e9 * Cm
    port.close()
    cases = [(st.n, np.asarray(st.rowptr), np.asarray(st.colidx), J)]
    for n in (20, 70):
        A = sp.random(n, n, density=min(0.5, 4.0 / n), random_state=int(rng.integers(1 << 30)), format="lil")
        for i in range(n):
            A[i, i] = 4.0 + rng.random()
            A[i, (i + 1) % n] = rng.random() - 0.5
            A[(i + 3) % n, i] = rng.random() - 0.5
        A = A.tocsr(); A.sort_indices()
        cases.append((n, A.indptr, A.indices, A.data))
    for n, rp, ci, vals in cases:
        if nc > n:
            continue
        P = hip.host_lu_analyze(n, rp, ci, vals, f2_nc=nc)
        rhs = rng.random(n) - 0.5
        x, (n_pre, n_post) = _emulate_f2(P, n, vals, rp, ci, rhs)
        A = sp.csr_matrix((vals, ci, rp), shape=(n, n))
        ref = spla.spsolve(A.tocsc(), rhs)
        bw = np.max(np.abs(A @ x - rhs) / (abs(A) @ np.abs(x) + np.abs(rhs) + 1e-300))
        assert bw < 1e-9, (n, nc, bw)
        assert np.max(np.abs(x - ref)) <= 1e-6 * max(1.0, np.max(np.abs(ref))), (n, nc)
        if n == st.n:
            assert n_pre + n_post <= (30 if nc == 0 else 19), (nc, n_pre, n_post)   # 29 passes without a core, 17-18 with


def _run_steps(W, words, nw, three, s0, cnt):
    """Execute steps [s0, s0 + cnt) of a straight-line step list on the work array W, lane by lane as the kernels do (fused2_kernel.hpp:
    run_steps -- 16-byte descriptors, three terms per lane; fused_team_kernel.hpp / lu_f2.hip: k_lu_steps -- the same for teams, and the
    8-byte one-term team layout): every lane reads its operands, the lane groups are summed, the leaders store."""
    NT = 64 * nw
    fld = lambda x, sh: ((x >> np.uint64(sh)) & np.uint64(0x7FFF)).astype(np.int64)
    bit = lambda x, sh: ((x >> np.uint64(sh)) & np.uint64(1)).astype(np.int64)
    d = words.reshape(-1, 2) if three else words.reshape(-1, 1)
    for s in range(s0, s0 + cnt):
        D = d[s * NT:(s + 1) * NT]
        lo = D[:, 0]
        pos, piv, a0, b0 = fld(lo, 0), fld(lo, 16), fld(lo, 32), fld(lo, 48)
        leader = bit(lo, 15)
        lg = bit(lo, 31) | bit(lo, 47) << 1 | bit(lo, 63) << 2
        part = W[a0] * W[b0]
        if three:
            hi = D[:, 1]
            part = part + W[fld(hi, 0)] * W[fld(hi, 16)] + W[fld(hi, 32)] * W[fld(hi, 48)]
            maxlg = bit(hi, 15) | bit(hi, 31) << 1 | bit(hi, 47) << 2
            hasdiv = bit(hi, 63)
            assert len(set(maxlg.tolist())) == 1 and len(set(hasdiv.tolist())) == 1 and lg.max() <= maxlg[0]     # the step's flags: the same in every lane
            if not hasdiv[0]:
                assert np.all(W[piv] == 1.0)
        grp = np.array([part[(l >> lg[l]) << lg[l]:((l >> lg[l]) << lg[l]) + (1 << lg[l])].sum() for l in range(NT)])
        acc = (W[pos] - grp) / W[piv]
        assert len(set(pos[leader == 1].tolist())) == int(leader.sum())       # one writer per word and step
        W[pos[leader == 1]] = acc[leader == 1]


@pytest.mark.parametrize("nc", [0, 8, 12])
def test_step_programs_on_cpu(nc):
    """The linear solve as straight-line steps (csrc/f2_program.cpp): f2_build_steps -- list-scheduled, three terms per lane: the fused sweep
    kernel's lean variant (one wave) and the per-op step LU (four waves) -- and f2_build_team (level-aligned, one term per lane: the team
    kernel), executed step by step on the DFF Jacobian and a random MNA-like matrix against SuperLU; the DFF's step counts are pinned (the
    LDS budget of the sweep kernel depends on them)."""
    import scipy.sparse.linalg as spla
    st, port = make_port(bm.dff_circuit(), {"vdd": 5.0})
    rng = np.random.default_rng(11)
    u = rng.random(st.n) * 5.0
    G, Cm, b, lw = port.rebuild(u, 2.005e-7)
    J = G + 1e9 * Cm
    port.close()
    cases = [(st.n, np.asarray(st.rowptr), np.asarray(st.colidx), J, hip.leaves_of(st))]
    n = 70
    A = sp.random(n, n, density=4.0 / n, random_state=5, format="lil")
    for i in range(n):
        A[i, i] = 4.0 + rng.random(); A[i, (i + 1) % n] = rng.random() - 0.5; A[(i + 3) % n, i] = rng.random() - 0.5
    A = A.tocsr(); A.sort_indices()
    cases.append((n, A.indptr, A.indices, A.data, None))
    for n, rp, ci, vals, leaves in cases:
        if nc > n:
            continue
        P = hip.host_lu_analyze(n, rp, ci, vals, f2_nc=nc, leaves=leaves)
        f2 = P["f2"]
        ncc, lu_words, dn0 = [int(x) for x in f2["meta"][:3]]
        y0 = lu_words
        pinv = np.empty(n, int); pinv[P["rperm"]] = np.arange(n)
        dst = np.zeros(len(vals), int); dst[P["load_src"]] = f2["posW"][P["load_dst"]]
        Asp = sp.csr_matrix((vals, ci, rp), shape=(n, n))
        rhs = rng.random(n) - 0.5
        for key, nw, three in ((1, 1, True), (14, 4, True), (2, 2, False), (4, 4, False)):
            (n_pre, n_post, n_fwd), words = P["steps"][key]
            W = np.zeros(y0 + n + 64 + 2); W[y0 + n + 64 + 1] = 1.0
            W[dst] = vals
            W[y0 + pinv] = rhs
            _run_steps(W, words, nw, three, 0, n_pre)
            if ncc:
                S = W[dn0:dn0 + ncc * ncc].reshape(ncc, ncc); yc = slice(y0 + n - ncc, y0 + n)
                W[yc] = np.linalg.solve(S, W[yc])
            _run_steps(W, words, nw, three, n_pre, n_post)
            x = np.empty(n); x[P["cperm"]] = W[y0:y0 + n]
            bw = np.max(np.abs(Asp @ x - rhs) / (abs(Asp) @ np.abs(x) + np.abs(rhs) + 1e-300))
            assert bw < 1e-9, (n, nc, key, bw)
            if n == st.n and nc == 8:
                assert (n_pre, n_post, n_fwd) == {1: (7, 6, 5), 14: (7, 5, 5), 2: (8, 9, 5), 4: (7, 6, 5)}[key], (key, n_pre, n_post, n_fwd)


def test_klu_style_ordering_on_cpu():
    """csrc/klu_order.cpp (CADNIP_LU_ORDER=klu; the reference's KLU orders with BTF + AMD, src/sweeps.jl:600): a randomly permuted block upper
    triangular matrix is recognised block for block (maximum transversal + strongly connected components), a matrix with a structurally
    zero diagonal gets a perfect matching, and the entry program built on the order (pivot rows by threshold partial pivoting on the sample,
    the matched entry preferred) solves random MNA-like systems and the flip-flop's Jacobian to a 1e-9 backward error."""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(5)
    sizes = [7, 1, 12, 1, 1, 20, 5]
    n = sum(sizes)
    A = sp.lil_matrix((n, n))
    o = 0
    for sz in sizes:
        for i in range(sz):                                     # an irreducible diagonal block: a cycle plus random entries
            A[o + i, o + i] = 4.0 + rng.random()
            A[o + i, o + (i + 1) % sz] = rng.random() + 0.1
            for j in rng.integers(0, sz, 2):
                A[o + i, o + j] = A[o + i, o + j] or rng.random() + 0.1
        for i in range(sz):                                     # coupling to LATER blocks only
            for j in rng.integers(o + sz, n, 2) if o + sz < n else []:
                A[o + i, j] = rng.random()
        o += sz
    pr, pc = rng.permutation(n), rng.permutation(n)
    Ap = A.tocsr()[pr][:, pc].tocsr(); Ap.sort_indices()
    P = hip.host_lu_analyze(n, Ap.indptr, Ap.indices, Ap.data, order="klu")
    assert P["n_blocks"] == len(sizes)
    # (zero diagonal: a voltage source's branch row has no diagonal entry, devices.jl:619-633)
    st, port = make_port(bm.dff_circuit(), {"vdd": 5.0})
    u = np.random.default_rng(11).random(st.n) * 5.0            # (the state test_fused_linear_solve_program_on_cpu uses)
    G, Cm, b, lw = port.rebuild(u, 2.005e-7)
    J = G + 1e9 * Cm
    port.close()
    cases = [(st.n, np.asarray(st.rowptr), np.asarray(st.colidx), J), (n, Ap.indptr, Ap.indices, Ap.data)]
    for nn, rp, ci, vals in cases:
        P = hip.host_lu_analyze(nn, rp, ci, vals, f2_nc=8, order="klu")
        assert P["n_blocks"] >= 1
        rhs = rng.random(nn) - 0.5
        x, _ = _emulate_f2(P, nn, vals, rp, ci, rhs)
        Asp = sp.csr_matrix((vals, ci, rp), shape=(nn, nn))
        bw = np.max(np.abs(Asp @ x - rhs) / (abs(Asp) @ np.abs(x) + np.abs(rhs) + 1e-300))
        assert bw < 1e-9, (nn, bw)


def test_cpu_port_ring_oscillator_fixture():
    """test/mna/vadistiller_integration.jl:649-692: the 3-stage sp_mos1 ring oscillates -- swing > 2 V, max > 2.5 V,
    min < 0.8 V, more than 10 mid-level crossings between 100 and 200 ns (dtmax = 1 ns).  The reference starts from
    CedarUICOp (no DC point); here the start state is u = 0 with out1 at the rail: from the perfectly symmetric
    all-zero state the three identical stages stay identical (the exact solution has no oscillation to find)."""
    circ = tc.ring_oscillator()
    st, port = make_port(circ, {}, 27.0, "tran")
    analyze_port(st, port, 3.3)
    u0 = np.zeros(st.n)
    u0[st.index_of("vdd")] = 3.3
    u0[st.index_of("out1")] = 3.3
    ts = np.linspace(100e-9, 200e-9, 500)
    out, _, rst, _ = port.tran(u0, 0.0, 200e-9, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4, save_t=ts,
                               obs=[st.index_of("out1")], err_mask=st.differential_mask(), use_pcnr=False, hmax=1e-9)
    assert rst["status"] == 1
    tc.ring_checks(out[:, 0])
    port.close()


def test_spice_deck_reader_flattens_subcircuits():
    """.SUBCKT hierarchy -> flat device table with the reference's naming (instance m1 in x1 in xu1 -> xu1_x1_m1, internal net n
    -> xu1_x1_n; src/spc/codegen.jl:745-757, 1708-1715): the hierarchical deck and its hand-flattened twin give the same
    table, structure and sweep behaviour."""
    hier = """* buffer chain from inverter cells
    .param wbase=1u
    .global vdd
    .model nch nmos level=1 vto=0.7 kp=100e-6
    .model pch pmos level=1 vto=-0.7 kp=50e-6
    .subckt inv in out wn=1u ratio=2
    .param wp={wn*ratio}
    MP out in vdd vdd pch w={wp} l=1e-6
    MN out in 0 0 nch w={wn} l=1e-6
    Cl out 0 {2f*ratio}
    .ends
    .subckt buf a y PARAMS: w=1u
    Xi1 a mid inv wn={w}
    Xi2 mid y inv wn={2*w} ratio=3
    Rleak mid 0 1meg
    .ends inv
    Vdd vdd 0 DC supply
    Vin in 0 DC 0 PULSE(0 3.3 1n 1n 1n 5n 20n)
    Xb1 in n1 buf w={wbase}
    Xb2 n1 out buf w='wbase*1.5'
    Bmon mon 0 V=V(n1)-V(out)
    Rs out 0 {supply*1k + 500}
    .end
    """
    flat = """* the same, flattened by hand
    .model nch nmos level=1 vto=0.7 kp=100e-6
    .model pch pmos level=1 vto=-0.7 kp=50e-6
    Vdd vdd 0 DC supply
    Vin in 0 DC 0 PULSE(0 3.3 1n 1n 1n 5n 20n)
    MXb1_Xi1_MP Xb1_mid in vdd vdd pch w=2e-6 l=1e-6
    MXb1_Xi1_MN Xb1_mid in 0 0 nch w=1e-6 l=1e-6
    CXb1_Xi1_Cl Xb1_mid 0 4f
    MXb1_Xi2_MP n1 Xb1_mid vdd vdd pch w=6e-6 l=1e-6
    MXb1_Xi2_MN n1 Xb1_mid 0 0 nch w=2e-6 l=1e-6
    CXb1_Xi2_Cl n1 0 6f
    RXb1_Rleak Xb1_mid 0 1meg
    MXb2_Xi1_MP Xb2_mid n1 vdd vdd pch w=3e-6 l=1e-6
    MXb2_Xi1_MN Xb2_mid n1 0 0 nch w=1.5e-6 l=1e-6
    CXb2_Xi1_Cl Xb2_mid 0 4f
    MXb2_Xi2_MP out Xb2_mid vdd vdd pch w=9e-6 l=1e-6
    MXb2_Xi2_MN out Xb2_mid 0 0 nch w=3e-6 l=1e-6
    CXb2_Xi2_Cl out 0 6f
    RXb2_Rleak Xb2_mid 0 1meg
    Bmon mon 0 V=V(n1)-V(out)
    Rs out 0 3800
    .end
    """
    ch, _ = cj.netlist.read_spice(hier, sweep=("supply",))
    cf, _ = cj.netlist.read_spice(flat, sweep=("supply",))
    gh, gf = ch.to_dicts({"supply": 3.3}), cf.to_dicts({"supply": 3.3})
    # the flat deck's element names carry the SPICE type letter in front; everything else must agree
    assert [d["name"] for d in gh] == [d["name"][1:] if "_" in d["name"] else d["name"] for d in gf]
    for a, b in zip(gh, gf):
        assert a["type"] == b["type"] and a["nodes"] == b["nodes"], (a, b)
        for k in a:
            if k in ("name", "type", "nodes"):
                continue
            if isinstance(a[k], dict):
                assert a[k] == pytest.approx(b[k]), (a["name"], k)
            elif isinstance(a[k], float):
                assert a[k] == pytest.approx(b[k], rel=1e-14), (a["name"], k)
            else:
                assert a[k] == b[k], (a["name"], k)
    sa, sb = cj.discover(ch, {"supply": 3.3}), cj.discover(cf, {"supply": 3.3})
    assert sa.n == sb.n and sa.nnz == sb.nnz and sa.node_names == sb.node_names and np.array_equal(sa.colidx, sb.colidx)
    # the sweep parameter went through an affine expression
    rs = [d for d in ch.to_dicts({"supply": 2.0}) if d["name"] == "Rs"][0]
    assert rs["r"] == pytest.approx(2500.0)
    # error behaviour
    for bad, exc in ((".subckt a p\nR1 p 0 1\n.ends\nX1 n1 n2 a\n", ValueError),            # port count
                     (".subckt a p\nR1 p 0 1\n.ends\nX1 n1 a rr=2\n", ValueError),           # unknown call parameter
                     (".subckt a p\nX1 p a\n.ends\nX1 n1 a\n", ValueError),                  # recursion
                     (".subckt a p\nR1 p 0 {nope}\n.ends\nX1 n1 a\n", KeyError),             # unknown name in an expression
                     ("R1 a 0 {supply*supply}\n", ValueError)):                                 # not affine in the sweep parameter
        with pytest.raises(exc):
            cj.netlist.read_spice(bad, sweep=("supply",))
    assert cj.netlist.eval_expr("2*(1k+500)/3 + sqrt(16) - 2^3", lambda n: 0.0) == pytest.approx(996.0)
    # X cards naming a Verilog-A module of the library become VA instances, also from inside a .SUBCKT
    vd, _ = cj.netlist.read_spice(""".subckt clamp a k rser=25
    Xd a k va_diode rs={rser} cj0=2p
    Xr k 0 va_resistor r=1k m=2
    .ends
    V1 in 0 DC 1
    X1 in out clamp rser=50
    """)
    got = vd.to_dicts({})
    assert [(d["type"], d["name"], d["nodes"]) for d in got] == [("V", "V1", ["in", "0"]), ("VA:va_diode", "X1_Xd", ["in", "out"]), ("VA:va_resistor", "X1_Xr", ["out", "0"])]
    assert got[1]["model"] == {"rs": 50.0, "cj0": pytest.approx(2e-12)} and got[2]["m"] == 2.0 and cj.discover(vd, {}).n == 5


def test_verilog_a_front_end_and_generator():
    """cadnip.jl_amd/va: parsing, the static analysis the generators rely on (branches in first-use order, which locals are
    duals, which carry a ddt() part), host-side evaluation, the emitted HIP text, and the constructs that must be refused."""
    from cadnip_jl_amd import va
    from cadnip_jl_amd.va import hipgen
    src = """
    `include "disciplines.vams"
    module tst(a, b, c);
      inout a, b, c; electrical a, b, c, x;      // x: internal node
      parameter real g = 1m from (0:inf);
      parameter real cq = 2p;
      parameter real k = 2.0 * g;                 /* default from an earlier parameter */
      real v, i, q, w;
      analog begin
        v = V(a, x);  w = cq * 3.0;
        if (v > 0.5) i = g * v * v; else i = k * v;
        q = w * v * (1.0 + 0.1 * v);
        I(a, x) <+ i + ddt(q);
        I(x, b) <+ V(x, b) * g;
        I(c) <+ 1u * tanh(V(c)) - ddt(w * V(c, a));
        I(a, x) <+ $simparam("gmin", 1e-12) * V(a, x);
      end
    endmodule
    """
    m = va.parse_module(src)
    assert (m.name, m.ports, m.nodes) == ("tst", ["a", "b", "c"], ["a", "b", "c", "x"]) and list(m.params) == ["g", "cq", "k"]
    assert m.branches == [(0, 3), (3, 1), (2, -1)] and m.reactive == [True, False, True]
    assert m.var_is_dual == {"v": True, "i": True, "q": True, "w": False} and not any(m.var_is_reactive.values())
    assert m.shape() == (4 + 3, 2 * 4 * 3 + 5 * 3, 2 * 3 + 2 * 4 * 3, 9, 3 + 3, 3)      # (third ipar row: which V(a,b) <+ 0 statements execute)
    par = va.host_eval.defaults(m, {"G": 2e-3})
    assert par == {"g": 2e-3, "cq": 2e-12, "k": 4e-3}
    with pytest.raises(va.VAError):
        va.host_eval.defaults(m, {"nope": 1.0})
    (i0, q0), (i1, q1), (i2, q2) = va.host_eval.evaluate(m, [1.0, 0.2, -0.3, 0.25], par, gmin=1e-12)
    assert i0 == pytest.approx(2e-3 * 0.75 ** 2 + 1e-12 * 0.75) and q0 == pytest.approx(6e-12 * 0.75 * 1.075)
    assert i1 == pytest.approx(0.05 * 2e-3) and q1 == 0.0
    assert i2 == pytest.approx(1e-6 * np.tanh(-0.3)) and q2 == pytest.approx(-6e-12 * (-1.3))
    text = hipgen.generate_function(m)
    assert "stamp_va_tst" in text and "constexpr int N = 4, B = 3, S = 0, NL = 0;" in text
    assert "double v_w = 0.0;" in text and "T v_q = 0.0;" in text            # w never sees a voltage: stays a double
    assert "va_emit_branch<N, S, B, true>(d, u, s, Vf, ld, nd, 0, 0, 3," in text and "va_emit_branch<N, S, B, true>(d, u, s, Vf, ld, nd, 2, 2, -1," in text
    hdr = hipgen.generate_header([m])
    assert '{"tst", 7, 39, 30, 9, 6, 3}' in hdr and "case 0: stamp_va_tst(d, u, s, lw); break;" in hdr
    # a local that carries ddt(): the reactive part follows it through assignments, sums and scaling
    m2 = va.parse_module("module r(p, n); electrical p, n; parameter real c = 1p; real t1, t2;"
                         " analog begin t1 = ddt(c * V(p, n)); t2 = 2.0 * t1 + V(p, n) * 1m; I(p, n) <+ -t2 / 4.0; end endmodule")
    assert m2.var_is_reactive == {"t1": True, "t2": True} and m2.reactive == [True]
    (ir, qr), = va.host_eval.evaluate(m2, [0.8, 0.0], {"c": 1e-12})
    assert ir == pytest.approx(-0.8e-3 / 4.0) and qr == pytest.approx(-2.0 * 0.8e-12 / 4.0)
    # the modules that ship in the library parse, and their ids are their positions
    reg = va.registry()
    assert [reg[n][0] for n in ("va_resistor", "va_capacitor", "va_diode", "va_sqmos", "va_dlim")] == [0, 1, 2, 3, 4]
    # analog functions and $limit sites (vasim.jl:1258-1330): one limit unknown per probe branch, one dual slot per site
    m3 = va.parse_module("""module lim(a, c, e); electrical a, c, e;
      analog function real clip; input vnew, vold, k; real t; begin t = vnew; if (vnew > vold + k) t = vold + k; clip = t; end endfunction
      analog function real twice; input x; begin twice = 2.0 * clip(x, 0.0, 1.0); end endfunction
      real v1, v2, v3;
      analog begin
        v1 = $limit(V(a, c), clip, 0.1);  v2 = $limit(V(e), clip, 0.2);  v3 = $limit(V(a, c), clip, 0.3);
        I(a, c) <+ 1m * v1 * v3 + twice(V(e));  I(e) <+ 2m * v2;
      end endmodule""")
    assert m3.limit_branches == [(0, 1), (2, -1)] and m3.limit_sites == [0, 1, 0] and list(m3.functions) == ["clip", "twice"]
    assert m3.shape() == (3 + 2 + 2, 2 * 3 * 2 + 4 * 2 + 3 * 2, 2 * 2 + 2 * 3 * 2, 6, 3, 3)
    assert m3.program([False, False])[:5] == [("G", 20, 5, 5), ("G", 21, 5, 0), ("G", 22, 5, 1), ("G", 23, 6, 6), ("G", 24, 6, 2)]
    (ia, _), (ie, _) = va.host_eval.evaluate(m3, [1.0, 0.2, 0.7], {}, vold=[0.5, 0.1])
    assert ia == pytest.approx(1e-3 * 0.6 * 0.8 + 2.0 * 0.7) and ie == pytest.approx(2e-3 * 0.3)      # clipped at vold + k; twice() clips at 1.0
    t3 = hipgen.generate_function(m3)
    assert "vaf_lim_clip<double>(va_val((V0 - V1)), vold0, va_val(0.3), sys)" in t3 and "const T site2 = va_site((V0 - V1), lim_w2, N + 2);" in t3
    assert "typedef Dual<N + S> T;" in t3 and "S = 3, NL = 2" in t3 and "vaf_lim_twice<T>(V2, sys)" in t3
    # named branches are plain aliases of their net pair
    m4 = va.parse_module("module nb(a, b); electrical a, b; branch (a, b) ab, again; branch (b) bg; parameter real g = 1m;"
                         " analog begin I(ab) <+ g * V(again); I(bg) <+ 2.0 * g * V(bg); end endmodule")
    assert m4.branches == [(0, 1), (1, -1)]
    assert va.host_eval.evaluate(m4, [1.0, 0.25], {"g": 1e-3}) == [(pytest.approx(0.75e-3), 0.0), (pytest.approx(0.5e-3), 0.0)]
    # node collapse: V(a,b) <+ 0 under a parameter-only condition aliases the internal node for the instances it holds for
    md = va.get("va_diode")[1]
    assert md.shorts and md.aliases(va.host_eval.defaults(md, {"rs": 0.0})) == {2: 0} and md.aliases(va.host_eval.defaults(md, {})) == {}
    c1, c2 = cj.Circuit(), cj.Circuit()
    for cc, rs in ((c1, 0.0), (c2, 5.0)):
        cc.V("v", "a", "0", dc=1.0)
        cc.VA("x", "va_diode", ("a", "0"), rs=rs)
    assert cj.discover(c1, {}).node_names == ["a"] and cj.discover(c2, {}).node_names == ["a", "x_va_diode_ai"]
    # refused, never approximated
    ok = va.parse_module("module x(a); electrical a; analog V(a) <+ 1.0; endmodule")             # a potential contribution with a value: its own branch current
    assert ok.vshorts == [0] and ok.short_kind == ["top"] and ok.shape()[0] == 1 + 0 + 0 + 1
    for bad in ("module x(a); electrical a; analog V(a) <+ ddt(V(a)); endmodule",                 # ddt() in a two-node potential contribution
                "module x(a, b); electrical a, b; analog if (V(a) > 0) V(a, b) <+ 0; endmodule",  # collapse under a voltage condition
                'module x(a); electrical a; real v; analog begin v = $limit(V(a), "pnjlim", 1, 2); I(a) <+ v; end endmodule',   # string form
                "module x(a); electrical a; real v; analog begin v = $limit(V(a), nofn, 1.0); I(a) <+ v; end endmodule",
                "module x(a); electrical a; analog function real f; input p, q; begin f = p; end endfunction real v;"
                " analog begin if (V(a) > 0) v = $limit(V(a), f); I(a) <+ v; end endmodule",        # $limit under a conditional
                "module x(a); electrical a; analog function real f; input p; begin f = f(p); end endfunction analog I(a) <+ f(V(a)); endmodule",
                "module x(a); electrical a; analog I(a) <+ ddt(V(a)) * ddt(V(a)); endmodule",     # product of two ddt()
                "module x(a); electrical a; analog I(a) <+ exp(ddt(V(a))); endmodule",            # ddt inside a function
                "module x(a); electrical a; analog I(a) <+ undeclared * V(a); endmodule",
                "module x(a); electrical a; analog I(a, b) <+ V(a); endmodule",                    # unknown net
                "module x(a); electrical a; analog @(cross(V(a), 1)) I(a) <+ 1; endmodule",
                "module x(a); analog I(a) <+ 1; endmodule"):                                       # port not electrical
        with pytest.raises(va.VAError):
            va.parse_module(bad)


def test_c6288_deck_flattens_and_orders_at_scale():
    """Deck reader and host symbolic phase on a single large circuit (c6288: three levels of .SUBCKT, .GLOBAL supplies,
    10 112 MOSFETs): device count, unknown layout, and the restricted Markowitz order (symbolic.cpp: beyond 4 096 unknowns
    only the shortest rows are searched) keeps the fill below 20 % on a diagonally dominant sample."""
    from tools.c6288 import deck
    circ = deck()
    # voltage sources first (codegen.jl:3130-3149), then the gates in netlist order, hierarchical names joined with "_"
    assert sum(d.type == "MOS1" for d in circ.devices) == 10112 and circ.devices[33].name == "vb15"
    assert circ.devices[34].name == "x1_xAND2_1_xmp2" and circ.devices[-1].name == "x1_xNOR2_2416_xmn2"
    st = cj.discover(circ, {})
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits) == (75908, 5090, 34, 30336, 40448)
    rng = np.random.default_rng(0)
    vals = rng.random(st.nnz) + 0.5
    rows = np.repeat(np.arange(st.n), np.diff(st.rowptr))
    vals[rows == st.colidx] += 10.0
    prog = hip.host_lu_analyze(st.n, st.rowptr, st.colidx, vals, sample=True)
    assert sorted(prog["rperm"].tolist()) == list(range(st.n)) and sorted(prog["cperm"].tolist()) == list(range(st.n))
    assert len(prog["col"]) < 1.2 * st.nnz
    # beyond 4 096 unknowns the default is KLU's ordering (csrc/klu_order.cpp: block triangular form, minimum degree inside the blocks): the
    # multiplier falls into tens of thousands of blocks (every charge / limit unknown is one), and against the restricted Markowitz search the
    # program is shallower and does less than half the multiply-adds
    mk = hip.host_lu_analyze(st.n, st.rowptr, st.colidx, vals, sample=True, order="markowitz")
    assert prog["n_blocks"] > 10000 and mk["n_blocks"] == 0
    assert len(prog["lev_ptr"]) < 0.85 * len(mk["lev_ptr"]) and len(prog["term_a"]) < 0.5 * len(mk["term_a"]) and len(prog["col"]) <= len(mk["col"])


def test_staged_continuation_order_and_structure_classes():
    """dc!(cs) continuation as batch stages (api.continuation_stages): every index exactly once, each stage's points lie midway
    between solved ones, seeds prefer the converged point below; and a sweep that moves a model parameter across a structural
    boundary (rd = 0 collapses sp_mos1's internal drain node, mos1.va:716-721) is split into one class per structure."""
    from cadnip_jl_amd import api
    for n in (1, 2, 3, 7, 40, 64, 1000):
        stages = api.continuation_stages(n)
        flat = [i for st in stages for i in st]
        assert sorted(flat) == list(range(n)) and stages[0] == [0] and len(stages) <= 2 + int(np.ceil(np.log2(max(n, 1))))
        solved = {0}
        for st in stages[1:]:
            for i in st:
                assert any(j < i for j in solved)                 # a solved neighbour below exists (the reference's direction)
            solved |= set(st)
    assert api.seed_for(5, [0, 4, 8]) == 4 and api.seed_for(2, [4, 8]) == 4 and api.seed_for(3, []) is None
    c = cj.Circuit("rd sweep")
    c.V("vd", "d", "0", dc=2.0)
    c.V("vg", "g", "0", dc=cj.Param("vg"))
    c.MOS1("m1", "d", "g", "0", "0", dict(type=1, vto=0.7, kp=100e-6, rd=cj.Param("rd")), w=10e-6, l=1e-6)
    mc = api.MNACircuit(c, {"rd": 0.0, "vg": 1.5})
    pts = list(api.ProductSweep(api.Sweep(vg=[1.0, 1.5, 2.0]), api.Sweep(rd=[0.0, 10.0, 0.0, 20.0])))
    classes = api.structure_classes(mc, pts)
    assert len(classes) == 2
    (i0, st0), (i1, st1) = classes
    assert sorted(i0 + i1) == list(range(12)) and all(pts[i]["rd"] == 0.0 for i in i0) and all(pts[i]["rd"] != 0.0 for i in i1)
    assert st1.n == st0.n + 1 and "m1_sp_mos1_d_int" in st1.node_names and "m1_sp_mos1_d_int" not in st0.node_names
    assert st0.signature() != st1.signature() and st0.signature() == cj.discover(c, {"rd": 0.0, "vg": 7.0}).signature()


@pytest.mark.parametrize("name", ["dff", "linear_zoo", "diode", "nonlinear_zoo", "behavioral", "inverter", "mos1_rd", "va_zoo"])
def test_exporter_twin_from_the_reference_compiled_structure(name):
    """SURVEY.md 8f-1 without Julia: cadnip.jl_amd/export_twin.py is julia/CadnipHIP.jl's export_structure statement for statement.  Fed with
    what the reference's CompiledStructure holds -- here the oracle's (oracle/mna_ref.py: compile_structure, precompile.jl:312-443):
    colptr / rowval, the positional maps G_coo_to_idx / C_coo_to_idx, b_deferred_resolved -- and the device table in builder order, it
    must produce the arrays the product's own structure discovery produces: CSR pattern, permutation to the reference's nzval order,
    gather lists in COO order, b lists.  A table that does not account for every stamp of the builder pass is refused."""
    from cadnip_jl_amd import export_twin as X
    from oracle import mna_ref as M
    from oracle.netlist_ref import make_builder
    from tests import circuits as tc
    mk, params = tc.ALL_STAMP[name]
    circ = mk()
    st = cj.discover(circ, params)
    bld = make_builder(circ.to_dicts(params))
    spec = M.MNASpec(mode="tran")
    ctx = M.build_with_detection(bld, {}, spec)
    cs = M.compile_structure(bld, {}, spec, ctx=ctx)
    table = X.device_table(st)
    ex = X.export_structure(cs, table)
    for key in ("rowptr", "colidx", "to_ref_nz", "g_ptr", "g_slots", "c_ptr", "c_slots", "b_ptr", "b_slots", "diag_nz"):
        assert np.array_equal(ex[key], getattr(st, key)), (name, key)
    assert ex["ns"] == (st.ns_g, st.ns_c, st.ns_b) and len(ex["blocks"]) == len(st.blocks)
    for a, b in zip(ex["blocks"], st.blocks):
        assert a["type"] == b.type and a["count"] == b.count and np.array_equal(a["nodes"], b.nodes) and (a["g_base"], a["c_base"], a["b_base"]) == (b.g_base, b.c_base, b.b_base)
    short = list(table)
    short[-1] = X.DeviceRow(short[-1].type, short[-1].nodes, short[-1].ipar, short[-1].program[:-1], short[-1].shape)
    with pytest.raises(ValueError):
        X.export_structure(cs, short)


def test_port_external_stamper_reproduces_its_own_transient():
    """oracle/cpu_port.py: Port.set_stamper (the hook through which the literal Verilog-A interpreter drives the port's transient controller for
    circuits of generated models, tools/make_tran_fixtures.py): with the stamps of a second, ordinary port behind the callback the
    transient is the same doubles as the ordinary port's own."""
    from cadnip_jl_amd import benchmarks as bm
    from cadnip_jl_amd.structure import expand_breakpoints
    from tests.port_util import make_port, analyze_port
    circ = bm.inverter_circuit()
    st, a = make_port(circ, {"vdd": 5.0}, 27.0, "tran")
    _, b = make_port(circ, {"vdd": 5.0}, 27.0, "tran")
    _, helper = make_port(circ, {"vdd": 5.0}, 27.0, "tran")

    def stamper(u, t):
        G, C, bb, lw = helper.rebuild(u, t)
        full = np.zeros(st.n)
        full[st.n - st.n_limits:] = lw
        return G, C, bb, full
    prog = analyze_port(st, a, 5.0)
    b.set_lu(prog)
    b.set_stamper(stamper)
    a.set_spec(mode="tranop"); u0, ok, _ = a.dc(abstol=1e-9); a.set_spec(mode="tran")
    assert ok
    ts = np.linspace(0.0, 1.5e-7, 16)
    kw = dict(breaks=expand_breakpoints(st.breakpoints, (0.0, 1.5e-7)), save_t=ts, err_mask=st.differential_mask(), use_pcnr=False, newton_mode=2)
    atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
    oa, _, sa, _ = a.tran(u0, 0.0, 1.5e-7, atol, 1e-4, **kw)
    ob, _, sb, _ = b.tran(u0, 0.0, 1.5e-7, atol, 1e-4, **kw)
    assert sa["status"] == 1 and np.array_equal(oa, ob) and (sa["newton_iters"], sa["accepted"], sa["rejected"]) == (sb["newton_iters"], sb["accepted"], sb["rejected"])
    for p in (a, b, helper):
        p.close()
