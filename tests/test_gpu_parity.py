"""GPU parity: HIP path (through the C ABI) vs the oracle on the same seeded inputs."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import cadnip_jl_amd as cj
from cadnip_jl_amd import hip
from cadnip_jl_amd import api as api_mod
from oracle import mna_ref as M
from oracle.netlist_ref import make_builder
from tests.circuits import ALL_STAMP

pytestmark = pytest.mark.gpu

RTOL = 1e-12    # fp64 parity tolerance on stamped values (relative to the entry scale)


def _oracle(circ, params, mode, temp=27.0):
    b = make_builder(circ.to_dicts(params))
    spec = M.MNASpec(mode=mode, temp=temp)
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    return cs, ws


def _handle(circ, params, B=1, temps=27.0, mode="tran"):
    st = cj.discover(circ, params)
    h = hip.Handle(st, B)
    pp = {k: np.full(B, float(v)) for k, v in params.items()}
    h.set_params(cj.pack_params(st, circ, pp, np.full(B, temps) if np.isscalar(temps) else temps, B))
    h.set_spec(mode=mode)
    return st, h


def _close(a, b, scale=None):
    scale = max(np.max(np.abs(b)), 1e-300) if scale is None else scale
    return np.max(np.abs(a - b)) <= RTOL * scale + 1e-300


@pytest.mark.parametrize("name", list(ALL_STAMP))
@pytest.mark.parametrize("mode,t", [("tran", 0.0), ("tran", 1.3e-3), ("dcop", 0.0)])
def test_rebuild_matches_oracle(name, mode, t):
    mk, params = ALL_STAMP[name]
    circ = mk()
    if name == "dff" and t > 0:
        t = 2.005e-7
    cs, ws = _oracle(circ, params, mode)
    st, h = _handle(circ, params, mode=mode)
    assert st.n == cs.n and st.nnz == cs.G.nnz
    rng = np.random.default_rng(42)
    for trial in range(3):
        u = (rng.random(st.n) * 2 - 0.5) * (1.0 if trial else 0.0)
        if name in ("dff", "inverter") and trial == 2:
            u = rng.random(st.n) * 5.0
        M.fast_rebuild(ws, u, t)
        h.rebuild(u, t)
        G, C, b, lw = h.get_GCb()
        for got, ref in ((G[0], cs.G.data), (C[0], cs.C.data), (b[0], ws.dctx.b)):
            # entries are sums of stamps: compare relative to the largest stamp magnitude in the array
            assert _close(got, ref), (name, trial, np.max(np.abs(got - ref)), np.max(np.abs(ref)))
        if st.n_limits:
            assert _close(lw[0], ws.dctx.limit_w, scale=max(1.0, np.max(np.abs(ws.dctx.limit_w))))
        du = rng.random(st.n)
        r = h.residual(du, u)[0]
        rr = cs.C @ du + cs.G @ u - ws.dctx.b
        assert _close(r, rr, scale=max(np.max(np.abs(cs.G.data)) * max(1.0, np.max(np.abs(u))), np.max(np.abs(rr)), 1e-30))
        J = h.jacobian(1e7)[0]
        assert _close(J, cs.G.data + 1e7 * cs.C.data)
        if st.n <= 300:     # the dense form (precompile.jl:588-603; test/mna/audio_integration.jl:505-520 drives the boundary through it)
            Jd = h.jacobian_dense(1e7)[0]
            ref = (cs.G + 1e7 * cs.C).toarray()
            assert Jd.shape == ref.shape and _close(Jd, ref, scale=max(np.max(np.abs(ref)), 1e-30))
        # ODE form (solve.jl:2241-2276): du = b - G u, J = -G, each with its own restamp
        du_ref, J_ref = np.empty(st.n), np.empty(st.nnz)
        M.ode_rhs(du_ref, u, ws, t)
        M.ode_jac(J_ref, u, ws, t)
        assert _close(h.ode_rhs(u, t)[0], du_ref, scale=max(np.max(np.abs(cs.G.data)) * max(1.0, np.max(np.abs(u))), np.max(np.abs(du_ref)), 1e-30))
        assert _close(h.ode_jacobian(u, t)[0], J_ref, scale=max(np.max(np.abs(cs.G.data)), 1e-30))
    h.close()


@pytest.mark.parametrize("B", [16, 37, 150])
def test_rebuild_of_a_batch_matches_oracle(B):
    """The stamping kernels' launch geometry depends on the batch (csrc/stamp_csr.hip: instances per wave for small device types, tiles per
    chunk, grid): a batch of B flip-flop corners with their own supplies, temperatures, states and times -- B not a multiple of anything --
    restamped in one call; several instances against the oracle at 1e-12, and every instance against its own single-instance restamp bit
    for bit (the per-op path has one writer per word and sums in the reference's COO order: no batch dependence at all)."""
    mk, params = ALL_STAMP["dff"]
    circ = mk()
    rng = np.random.default_rng(B)
    vdds, temps = 4.5 + rng.random(B), -40.0 + 165.0 * rng.random(B)
    st = cj.discover(circ, params)
    h = hip.Handle(st, B)
    h.set_params(cj.pack_params(st, circ, {"vdd": vdds}, temps, B))
    h.set_spec(mode="tran")
    u = rng.random((B, st.n)) * 5.0
    t = rng.random(B) * 7e-7
    h.rebuild(u, t)
    G, C, b, lw = h.get_GCb()
    r = h.residual(u * 0.5, u)
    h.close()
    for i in sorted({0, 7, B // 2, B - 1}):
        cs, ws = _oracle(circ, {"vdd": float(vdds[i])}, "tran", temp=float(temps[i]))
        M.fast_rebuild(ws, u[i], float(t[i]))
        for got, ref in ((G[i], cs.G.data), (C[i], cs.C.data), (b[i], ws.dctx.b)):
            assert _close(got, ref), (B, i, np.max(np.abs(got - ref)), np.max(np.abs(ref)))
        st1, h1 = _handle(circ, {"vdd": float(vdds[i])}, temps=float(temps[i]))
        h1.rebuild(u[i], float(t[i]))
        G1, C1, b1, lw1 = h1.get_GCb()
        r1 = h1.residual(u[i] * 0.5, u[i])
        h1.close()
        assert np.array_equal(G1[0], G[i]) and np.array_equal(C1[0], C[i]) and np.array_equal(b1[0], b[i]) and np.array_equal(lw1[0], lw[i]) and np.array_equal(r1[0], r[i]), (B, i)


def test_initjct_stamp_matches_oracle():
    mk, params = ALL_STAMP["dff"]
    circ = mk()
    cs, ws = _oracle(circ, params, "tranop")
    st, h = _handle(circ, params, mode="tranop")
    u = np.zeros(st.n)
    ws.dctx.initjct = True
    M.fast_rebuild(ws, u, 0.0)
    ws.dctx.initjct = False
    h.set_initjct(True)
    h.rebuild(u, 0.0)
    h.set_initjct(False)
    G, C, b, lw = h.get_GCb()
    assert _close(G[0], cs.G.data) and _close(C[0], cs.C.data) and _close(b[0], ws.dctx.b)
    assert _close(lw[0], ws.dctx.limit_w, scale=5.0)
    h.close()


def test_batched_instances_are_independent():
    """B instances with different Vdd / temperature == B single-instance runs."""
    mk, _ = ALL_STAMP["dff"]
    circ = mk()
    vdds = np.array([4.5, 5.0, 5.5, 5.2])
    temps = np.array([-40.0, 27.0, 125.0, 85.0])
    st = cj.discover(circ, {"vdd": 5.0})
    h = hip.Handle(st, 4)
    h.set_params(cj.pack_params(st, circ, {"vdd": vdds}, temps, 4))
    h.set_spec(mode="tran")
    rng = np.random.default_rng(7)
    u = rng.random((4, st.n)) * 5
    tt = np.array([0.0, 5.1e-8, 2.005e-7, 6e-7])
    h.rebuild(u, tt)
    G, C, b, lw = h.get_GCb()
    for i in range(4):
        cs, ws = _oracle(circ, {"vdd": vdds[i]}, "tran", temp=temps[i])
        M.fast_rebuild(ws, u[i], tt[i])
        assert _close(G[i], cs.G.data) and _close(C[i], cs.C.data) and _close(b[i], ws.dctx.b)
    h.close()


@pytest.mark.parametrize("name", ["linear_zoo", "inverter", "dff", "mos1_rd"])
def test_lu_factor_solve(name):
    mk, params = ALL_STAMP[name]
    circ = mk()
    st, h = _handle(circ, params, B=3)
    rng = np.random.default_rng(3)
    u = rng.random((3, st.n)) * 3
    h.rebuild(u, 1e-7)
    gam = np.array([1e6, 1e8, 1e10])
    J = h.jacobian(gam)
    h.analyze_values(np.max(np.abs(J), axis=0))
    h.factor()
    rhs = rng.random((3, st.n))
    x = h.solve(rhs)
    for i in range(3):
        A = sp.csc_matrix((J[i], st.ref_rowval, st.ref_colptr), shape=(st.n, st.n))
        xr = spla.splu(A).solve(rhs[i])
        if name != "dff":   # random states make the DFF Jacobian numerically singular: only the backward error is meaningful
            assert np.max(np.abs(x[i] - xr)) <= 1e-6 * max(1.0, np.max(np.abs(xr))), (name, i)
        # backward error relative to |A||x| (random states make J badly conditioned)
        assert np.max(np.abs(A @ x[i] - rhs[i]) / (abs(A) @ np.abs(x[i]) + np.abs(rhs[i]))) <= 1e-12
    print(name, h.lu_stats())
    h.close()


def test_lu_forward_error_at_physical_dff_states():
    """The DFF Jacobian is numerically singular at random states (test above: backward error only).  At the states the
    solver actually meets -- the DC operating point and a mid-transient state, with gamma = 1/h of real steps -- the
    solution itself must agree with SuperLU (forward error)."""
    from cadnip_jl_amd import benchmarks as bm
    from cadnip_jl_amd.structure import expand_breakpoints
    circ = bm.dff_circuit()
    sim = api_mod.BatchSimulator(api_mod.MNACircuit(circ, {"vdd": 5.0}), [{"vdd": 5.0}, {"vdd": 4.6, "temp": 110.0}, {"vdd": 5.4, "temp": -20.0}])
    st, h = sim.st, sim.h
    sim.analyze()
    u_dc, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
    assert np.all(conv)
    h.set_spec(mode="tran")
    atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
    out, _, stats = h.tran_run(0.0, 2.05e-7, atol, 1e-4, breaks=expand_breakpoints(st.breakpoints, (0.0, 2.05e-7)), save_t=[2.05e-7], fused=0)
    assert stats["n_failed"] == 0
    rng = np.random.default_rng(5)
    for u, t, gam in ((u_dc, 0.0, np.array([0.0, 1e8, 1e10])), (out[:, 0, :], 2.05e-7, np.array([1e7, 1e9, 1e11]))):
        h.rebuild(u, t)
        J = h.jacobian(gam)
        h.factor()
        rhs = rng.standard_normal((3, st.n))
        x = h.solve(rhs)
        for i in range(3):
            A = sp.csc_matrix((J[i], st.ref_rowval, st.ref_colptr), shape=(st.n, st.n))
            xr = spla.splu(A).solve(rhs[i])
            assert np.max(np.abs(x[i] - xr)) <= 1e-8 * max(1.0, np.max(np.abs(xr))), (t, i, np.max(np.abs(x[i] - xr)), np.max(np.abs(xr)))
            assert np.max(np.abs(A @ x[i] - rhs[i]) / (abs(A) @ np.abs(x[i]) + 1e-300)) <= 1e-12
    sim.close()


def test_singular_matrix_reports_status():
    c = cj.Circuit()
    c.I("i1", "a", "0", dc=1e-3)
    c.C("c1", "a", "0", 1e-9)    # no DC path: G is structurally singular in DC
    c.R("r1", "b", "0", 1.0)
    st, h = _handle(c, {})
    h.rebuild(np.zeros(st.n), 0.0)
    h.jacobian(0.0)
    with pytest.raises(hip.CadnipError):
        h.analyze()
    h.close()


def test_abi_rejects_malformed_structures_and_calls():
    """Host-side shape validation of the C ABI: what a wrong binding would pass must come back as CADNIP_BADARG /
    CADNIP_NOTREADY, never reach a kernel."""
    import copy
    from tests.circuits import divider, rc_charge
    st = cj.discover(divider(), {})
    for mutate in ("node_oob", "slot_overflow", "rowptr_short", "zero_instances", "bad_type"):
        s2 = copy.deepcopy(st)
        B = 1
        if mutate == "node_oob":
            s2.blocks[1].nodes[0, 0] = st.n + 3
        elif mutate == "slot_overflow":
            s2.blocks[1].g_base = st.ns_g
        elif mutate == "rowptr_short":
            s2.rowptr = np.array(st.rowptr, copy=True); s2.rowptr[-1] = st.nnz + 5
        elif mutate == "zero_instances":
            B = 0
        elif mutate == "bad_type":
            s2.blocks[0].type = "R"; s2.blocks[0].n_par = 0
            hip.TYPE_ID_BACKUP = dict(hip.TYPE_ID)
            hip.TYPE_ID["R"] = 99
        try:
            with pytest.raises(hip.CadnipError) as ei:
                hip.Handle(s2, B)
            assert ei.value.code == hip.BADARG, mutate
        finally:
            if mutate == "bad_type":
                hip.TYPE_ID.clear(); hip.TYPE_ID.update(hip.TYPE_ID_BACKUP)
    # calls in the wrong order / with the wrong sizes
    circ = rc_charge()
    st = cj.discover(circ, {})
    h = hip.Handle(st, 2)
    h.set_params(cj.pack_params(st, circ, {}, 27.0, 2))
    with pytest.raises(hip.CadnipError) as ei:            # LU before the symbolic phase
        h.factor()
    assert ei.value.code == hip.NOTREADY
    with pytest.raises(hip.CadnipError) as ei:            # fused transient before the symbolic phase
        h.tran_run(0.0, 1e-3, np.full(st.n, 1e-9), 1e-6, save_t=np.array([1e-3]), fused=1)
    assert ei.value.code == hip.NOTREADY
    with pytest.raises((hip.CadnipError, ValueError)):    # parameter block of the wrong shape
        h.set_params([np.zeros((2, 1, 1))])
    h.close()


def test_transient_edge_cases():
    """Empty save list, save points on both ends, a single instance and a batch whose size is not a multiple of the
    workgroup's eight instances, all through the fused kernel."""
    circ = cj.Circuit()
    circ.V("v1", "vin", "0", dc=0.0, wave=("pwl", [0.0, 1e-9], [0.0, 5.0]))
    circ.R("r1", "vin", "out", 1e3)
    circ.C("c1", "out", "0", 1e-6)
    for B in (1, 3, 11):
        sim = api_mod.BatchSimulator(api_mod.MNACircuit(circ, {}), [{} for _ in range(B)])
        st = sim.st
        out, per, stats = sim.tran((0.0, 2e-3), np.full(st.n, 1e-9), 1e-6, np.array([0.0, 1e-3, 2e-3]), obs=[st.index_of("out")], fused=1)
        assert stats["n_failed"] == 0 and out.shape == (B, 3, 1)
        assert np.allclose(out[:, 1, 0], 5.0 * (1 - np.exp(-1.0)), rtol=1e-3) and np.allclose(out[:, 0, 0], out[0, 0, 0])
        out2, per2, stats2 = sim.tran((0.0, 2e-3), np.full(st.n, 1e-9), 1e-6, np.array([]), obs=[st.index_of("out")], fused=1)
        assert stats2["n_failed"] == 0 and out2.shape[1] == 0 and np.array_equal(per2[:, :3], per[:, :3])
        sim.close()


@pytest.mark.parametrize("name", ["dff", "linear_zoo", "diode"])
def test_newton_step_in_one_call_equals_the_five_calls(name):
    """cadnip_newton_step (rebuild -> residual -> [jacobian -> factor] -> solve in one call, its launch sequence replayed as a HIP graph) gives
    the same doubles as the five entry points called one after the other: with and without a refactorisation, first call (plain
    launches), second call (graph capture), later calls (graph replay), and again after the handle's configuration changed (a stale
    graph must not be replayed)."""
    import cadnip_jl_amd as cj
    from cadnip_jl_amd import benchmarks as bm, hip
    from tests import circuits as tc
    if name == "dff":
        circ, params = bm.dff_circuit(), {"vdd": 5.0}
    else:
        mk, params = tc.ALL_STAMP[name]
        circ = mk()
    B = 3
    st = cj.discover(circ, params)
    h = hip.Handle(st, B)
    h.set_params(cj.pack_params(st, circ, {k: np.full(B, float(v)) for k, v in params.items()}, np.array([27.0, -20.0, 110.0]), B))
    h.set_spec(mode="tran")
    rng = np.random.default_rng(5)
    vs = 5.0 if name == "dff" else 1.0
    gam = np.array([1e9, 3e8, 2e9])

    def five(u, du, t, refresh):
        h.rebuild(u, t)
        r = h.residual(du, u)
        if refresh:
            h.jacobian(gam, readback=False)
            h.factor()
        return h.solve(r), r

    u0 = rng.random((B, st.n)) * vs
    h.rebuild(u0, np.zeros(B)); h.jacobian(gam, readback=True); h.analyze(0)
    for rep in range(5):
        u, du, t = rng.random((B, st.n)) * vs, rng.random((B, st.n)) * 1e6, rng.random(B) * 1e-7
        for refresh in (True, False):
            if rep == 3 and refresh:
                h.set_spec(mode="tran", gmin=1e-11)        # the configuration moves: graphs captured before are stale
            x5, r5 = five(u, du, t, refresh)
            x1, nrm, r1 = h.newton_step(u, du, gam if refresh else None, t, refresh=refresh, want_resid=True)
            assert np.array_equal(r1, r5) and np.array_equal(x1, x5), (name, rep, refresh)
            assert np.allclose(nrm, np.sqrt(np.sum(r5 * r5, axis=1)), rtol=1e-12)
    h.close()


def test_newton_step_in_the_fused_kernel_agrees_with_the_per_op_kernels():
    """cadnip_newton_step_fused: the same iteration (residual, refactorisation or kept factors, solve) in the team kernel's STEP mode --
    another summation order, so the agreement is to rounding: residual 1e-12 of its largest entry, Newton step 1e-8 of its largest entry
    (conditioning of J = G + 1e9 C); a circuit outside the team kernel's device set is refused (the caller takes cadnip_newton_step)."""
    import cadnip_jl_amd as cj
    from cadnip_jl_amd import benchmarks as bm, hip
    from tests import circuits as tc
    from cadnip_jl_amd import api
    circ = bm.dff_circuit()
    B = 5
    sim = api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), [{"vdd": float(v), "temp": float(tc_)} for v, tc_ in zip(np.linspace(4.5, 5.5, B), np.linspace(-40.0, 125.0, B))])
    st, h = sim.st, sim.h
    sim.analyze()
    udc, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
    assert np.all(conv)
    h.set_spec(mode="tran")
    rng = np.random.default_rng(11)
    gam = np.full(B, 1e9)
    for rep in range(3):
        # states near the operating point (a random state sends the exponentials of the junctions to 1e58: nothing to compare there)
        u, du, t = udc + 1e-3 * (rng.random((B, st.n)) - 0.5), (rng.random((B, st.n)) - 0.5) * 1e5, rng.random(B) * 1e-7
        for refresh in (True, False):
            x1, n1, r1 = h.newton_step(u, du, gam if refresh else None, t, refresh=refresh, want_resid=True)
            x2, n2, r2 = h.newton_step(u, du, gam if refresh else None, t, refresh=refresh, want_resid=True, fused=True)
            assert np.max(np.abs(r1 - r2)) <= 1e-12 * np.max(np.abs(r1)) and np.allclose(n1, n2, rtol=1e-12)
            assert np.max(np.abs(x1 - x2)) <= 1e-8 * np.max(np.abs(x1)), (rep, refresh, np.max(np.abs(x1 - x2)), np.max(np.abs(x1)))
    sim.close()
    mk, params = tc.ALL_STAMP["diode"]
    circ = mk()
    st = cj.discover(circ, params)
    h = hip.Handle(st, 1)
    h.set_params(cj.pack_params(st, circ, {}, np.array([27.0]), 1))
    h.set_spec(mode="tran")
    u = np.zeros((1, st.n))
    h.rebuild(u, 0.0); h.jacobian(np.array([1e6]), readback=True); h.analyze(0)
    with pytest.raises(hip.CadnipError):
        h.newton_step(u, u, 1e6, 0.0, fused=True)
    h.newton_step(u, u, 1e6, 0.0)
    h.close()


@pytest.mark.parametrize("name", ["dff", "linear_zoo"])
def test_handle_built_from_the_exported_structure_matches_the_oracle(name):
    """the structure a Julia host would export from the reference's CompiledStructure (cadnip.jl_amd/export_twin.py, the twin of
    julia/CadnipHIP.jl: export_structure, run on the oracle's CompiledStructure) builds a handle whose rebuild / residual / Jacobian equal the
    oracle's fast_rebuild! / fast_residual! / fast_jacobian! at 1e-12."""
    import cadnip_jl_amd as cj
    from cadnip_jl_amd import export_twin as X, hip
    from oracle import mna_ref as M
    from oracle.netlist_ref import make_builder
    from tests import circuits as tc
    mk, params = tc.ALL_STAMP[name]
    circ = mk()
    st0 = cj.discover(circ, params)
    bld = make_builder(circ.to_dicts(params))
    spec = M.MNASpec(mode="tran", temp=27.0)
    ctx = M.build_with_detection(bld, {}, spec)
    cs = M.compile_structure(bld, {}, spec, ctx=ctx)
    st = X.to_structure(X.export_structure(cs, X.device_table(st0)), st0)
    h = hip.Handle(st, 1)
    h.set_params(cj.pack_params(st, circ, {k: np.array([float(v)]) for k, v in params.items()}, np.array([27.0]), 1))
    h.set_spec(mode="tran")
    rng = np.random.default_rng(3)
    ws = M.create_workspace(cs, ctx=ctx)
    for rep in range(3):
        u, du, t, gamma = rng.random(st.n) * (5.0 if name == "dff" else 1.0), rng.random(st.n) * 1e3, 1e-8 * (rep + 1), 1e7
        h.rebuild(u[None, :], t)
        G, C, b, _ = h.get_GCb()
        M.fast_rebuild(ws, u, t)
        for got, ref in ((G[0], cs.G.data), (C[0], cs.C.data), (b[0], ws.dctx.b)):
            assert np.max(np.abs(got - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1e-300), (name, rep)
        r = h.residual(du[None, :], u[None, :])[0]
        ref_r = np.zeros(st.n)
        M.fast_residual(ref_r, du, u, ws, t)
        assert np.max(np.abs(r - ref_r)) <= 1e-12 * np.max(np.abs(ref_r))
        J = h.jacobian(np.array([gamma]))[0]
        assert np.max(np.abs(J - (cs.G.data + gamma * cs.C.data))) <= 1e-12 * np.max(np.abs(cs.G.data + gamma * cs.C.data))
    h.close()
