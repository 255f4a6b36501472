"""GPU: the DC fallback chain (_dc_solve_with_fallbacks, solve.jl:871-929: PCNR -> plain Newton -> gshunt stepping ->
source stepping) walked per sweep instance by cadnip_dc_run, against the oracle's literal restatement
(oracle/mna_ref.py:698-777) run with the same options: same ladder of gshunt / srcFact rungs, same Newton counts on every
rung, same verdict, solution within 1e-9."""
import numpy as np
import pytest

import cadnip_jl_amd as cj
from cadnip_jl_amd import api, hip
from oracle import mna_ref as M
from oracle.netlist_ref import make_builder
from tests import circuits as tc
from tests.dc_chain_util import oracle_chain, same_ladder

pytestmark = pytest.mark.gpu


def _oracle(circ, params=None, **kw):
    b = make_builder(circ.to_dicts(params or {}))
    spec = M.MNASpec(mode="dcop")
    ctx = M.build_with_detection(b, {}, spec)
    cs = M.compile_structure(b, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    return oracle_chain(cs, ws, np.zeros(cs.n), **kw)


def _gpu(circ, points=None, params=None, fused=False, **kw):
    sim = api.BatchSimulator(api.MNACircuit(circ, params or {}, api.MNASpec(mode="dcop")), points)
    try:
        sim.analyze()
        u, conv, stats = sim.h.dc_run(None, use_pcnr=True, cold_start=True, fused=fused, **kw)
        log = sim.h.dc_log()
        per = [[e[1:] for e in log if e[0] == i] for i in range(sim.B)]
        # the handle leaves the chain with its own spec: a plain restamp afterwards carries no homotopy terms
        sim.h.rebuild(u, 0.0)
        G0, _, b0, _ = sim.h.get_GCb()
        sim.h.set_spec(gshunt=0.0, srcFact=1.0)
        sim.h.rebuild(u, 0.0)
        G1, _, b1, _ = sim.h.get_GCb()
        assert np.array_equal(G0, G1) and np.array_equal(b0, b1)
        return u, conv, per, stats
    finally:
        sim.close()


CASES = {
    # plain Newton runs out of iterations (no limiting: the junction voltage comes down by ~Vt per iteration); the gshunt ladder arrives
    "rect_nolimit": (lambda: tc.diode_rectifier(False), {}, True, (1, 2)),
    # only the source ramp arrives (tight iteration budget)
    "bi_quadratic": (tc.bi_quadratic, dict(abstol=1e-7, maxiters=5), True, (1, 2, 3)),
    # every stage fails: PCNR out of iterations, the others on non-finite stamps
    "inverter_chain8": (lambda: tc.inverter_chain(8), dict(maxiters=12), False, (0, 1, 2, 3)),
    "rect_5_iterations": (lambda: tc.diode_rectifier(True), dict(maxiters=5), False, (0, 1, 2, 3)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_fallback_chain_matches_oracle_rung_for_rung(name):
    mk, kw, expect_ok, stages = CASES[name]
    u_ref, ok_ref, log_ref = _oracle(mk(), **kw)
    assert ok_ref == expect_ok and sorted({e[0] for e in log_ref}) == list(stages)
    u, conv, per, stats = _gpu(mk(), **kw)
    assert bool(conv[0]) == ok_ref
    assert same_ladder(per[0], log_ref), (per[0], log_ref)
    assert stats["newton_iters"] == sum(e[3] for e in log_ref)
    if ok_ref:
        assert np.max(np.abs(u[0] - u_ref)) <= 1e-9 * max(1.0, np.max(np.abs(u_ref)))


def _rectifier_sweep(limit):
    circ = cj.Circuit("rectifier sweep")
    circ.V("v1", "in", "0", dc=cj.Param("vin"))
    circ.R("r1", "in", "out", 1e3)
    circ.D("d1", "out", "0", Is=1e-14, limit=limit)
    return circ


def _decisive_prefix(log, maxiters):
    """Entries up to (excluding) the first rung that converged on its last or last-but-one allowed iteration: whether such a
    rung passes ||F|| < abstol is a matter of rounding, and everything after it depends on that verdict."""
    out = []
    for e in log:
        if e[2] and e[3] >= maxiters - 1:
            break
        out.append(e)
    return out


@pytest.mark.parametrize("fused", [False, True])
def test_batch_points_leave_the_chain_independently(fused):
    """Sweep points are independent circuits (sweeps.jl:696-703): each leaves the chain at the first stage that solves it.
    Rectifier with limiting and a 10-iteration budget: 5 V, 500 V, 3 V converge in PCNR (7-8 iterations); 0.7 V and 0.5 V
    need the plain-Newton stage (PCNR starts the junction at vcrit and limiting brings it down too slowly); 0.8 V needs the
    gshunt ladder.  Every ladder equals the oracle's, and nobody's result depends on who else is in the batch."""
    circ = _rectifier_sweep(True)
    vins = [5.0, 0.8, 500.0, 0.7, 3.0, 0.5]
    kw = dict(maxiters=10)
    ref = [_oracle(circ, {"vin": v}, **kw) for v in vins]
    assert all(r[1] for r in ref) and [max(e[0] for e in r[2]) for r in ref] == [0, 2, 0, 1, 0, 1]
    pts = [{"vin": v} for v in vins]
    u, conv, per, _ = _gpu(circ, pts, {"vin": 1.0}, fused=fused, **kw)
    assert np.all(conv)
    for i, (u_ref, ok_ref, log_ref) in enumerate(ref):
        assert same_ladder(per[i], log_ref), (i, per[i], log_ref)
        assert np.max(np.abs(u[i] - u_ref)) <= 1e-9 * max(1.0, np.max(np.abs(u_ref)))
    keep = [0, 2, 3, 4]                                     # without the two hardest points: bit-identical for the others
    u2, conv2, per2, _ = _gpu(circ, [pts[i] for i in keep], {"vin": 1.0}, fused=fused, **kw)
    assert np.all(conv2)
    for k, i in enumerate(keep):
        assert np.array_equal(u2[k], u[i]) and per2[k] == per[i]


def test_one_hopeless_point_does_not_disturb_the_batch():
    """The same rectifier without limiting, 100 iterations: 0.5 V and 2 V converge with plain Newton, 5 V on the gshunt ladder,
    8 V nowhere (plain Newton and every gshunt rung run out of iterations, the source ramp out of its 50 steps).  The hopeless
    point is reported as such, walks the oracle's ladder (compared up to the first rung whose verdict hangs on rounding), and
    the converged points are exactly what they are without it."""
    circ = _rectifier_sweep(False)
    vins = [0.5, 8.0, 5.0, 2.0]
    ref = [_oracle(circ, {"vin": v}) for v in vins]
    assert [r[1] for r in ref] == [True, False, True, True]
    pts = [{"vin": v} for v in vins]
    u, conv, per, _ = _gpu(circ, pts, {"vin": 1.0})
    assert list(conv) == [True, False, True, True]
    for i, (u_ref, ok_ref, log_ref) in enumerate(ref):
        if ok_ref:
            assert same_ladder(per[i], log_ref), (i, per[i], log_ref)
            assert np.max(np.abs(u[i] - u_ref)) <= 1e-9 * max(1.0, np.max(np.abs(u_ref)))
        else:
            want = _decisive_prefix(log_ref, 100)
            assert len(want) >= 8 and {e[0] for e in want} == {1, 2, 3}
            assert same_ladder(per[i][:len(want)], want), (per[i][:len(want)], want)
            assert {e[0] for e in per[i]} == {0, 2, 3} and sum(1 for e in per[i] if e[0] == 3) <= 50
    keep = [0, 2, 3]
    u2, conv2, per2, _ = _gpu(circ, [pts[i] for i in keep], {"vin": 1.0})
    assert np.all(conv2)
    for k, i in enumerate(keep):
        assert np.array_equal(u2[k], u[i]) and per2[k] == per[i]


def test_failed_reanalysis_keeps_the_previous_factorisation_usable():
    """A symbolic analysis that fails (singular sample -- what a re-pivot on a singular victim meets, driver.hip) must leave
    the handle's previous LU program and its device arrays in place: factor + solve still give the same answer."""
    mk, params = tc.ALL_STAMP["linear_zoo"]
    circ = mk()
    st = cj.discover(circ, params)
    h = hip.Handle(st, 2)
    h.set_params(cj.pack_params(st, circ, {}, np.full(2, 27.0), 2))
    rng = np.random.default_rng(11)
    h.rebuild(rng.random((2, st.n)), 0.0)
    J = h.jacobian(np.array([1e6, 1e8]))
    h.analyze_values(np.max(np.abs(J), axis=0))
    stats = h.lu_stats()
    h.factor()
    rhs = rng.random((2, st.n))
    x0 = h.solve(rhs)
    with pytest.raises(hip.CadnipError):
        h.analyze_values(np.zeros(st.nnz))                   # structurally singular sample
    assert h.lu_stats() == stats
    h.jacobian(np.array([1e6, 1e8]))
    h.factor()
    assert np.array_equal(h.solve(rhs), x0)
    h.close()


def test_rebuild_reports_nonfinite_stamps():
    """CADNIP_NONFINITE from cadnip_rebuild (the Julia shim maps it to DomainError, caught at solve.jl:887-897)."""
    circ = tc.diode_rectifier(False)
    st = cj.discover(circ, {})
    h = hip.Handle(st, 2)
    h.set_params(cj.pack_params(st, circ, {}, np.full(2, 27.0), 2))
    u = np.zeros((2, st.n))
    u[1, st.index_of("out")] = 40.0                          # exp(40 / 0.026) overflows: G, b of instance 1 are not finite
    with pytest.raises(hip.CadnipError) as ei:
        h.rebuild(u, 0.0)
    assert ei.value.code == hip.NONFINITE
    h.rebuild(np.zeros((2, st.n)), 0.0)                      # and the flag does not stick
    h.close()
