"""Transient parity at full benchmark size: GPU drivers vs the CPU port (same integrator policy).

north_star tolerance: node-voltage error <= 1e-9 relative.  The tolerance is applied to every recorded
unknown at every save time, relative to max(|value|, 1) (volts / scaled charges are O(1))."""
import numpy as np
import pytest

from cadnip_jl_amd import api, benchmarks as bm
from cadnip_jl_amd.structure import expand_breakpoints
from tests.port_util import make_port, analyze_port

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9
ABSTOL = dict(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)


def _port_run(circ, params, temp, u0, ts, obs, vscale):
    st, port = make_port(circ, params, temp, "tran")
    analyze_port(st, port, vscale)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    out, uf, stats, _ = port.tran(u0, bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], st.state_abstol(**ABSTOL), 1e-4, breaks=breaks, save_t=ts,
                                  obs=obs, err_mask=st.differential_mask(), use_pcnr=False)
    port.close()
    return out, stats


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("points", [[{}], [{"vdd": 4.5, "temp": -40.0}, {"vdd": 5.5, "temp": 125.0}, {"vdd": 4.5, "temp": 125.0}, {"vdd": 5.2, "temp": 60.0}]])
def test_dff_transient_matches_port(points, fused):
    circ = bm.dff_circuit()
    mc = api.MNACircuit(circ, {"vdd": 5.0})
    sim = api.BatchSimulator(mc, points)
    st = sim.st
    sim.analyze()
    u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
    assert np.all(conv)
    ts = np.linspace(0.0, 7e-7, 141)
    obs = list(range(st.n_nodes)) + [st.index_of("X_tn10_sp_mos1_Q_b_0")]
    sim.h.set_spec(mode="tran")
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    out, per, stats = sim.h.tran_run(0.0, 7e-7, st.state_abstol(**ABSTOL), 1e-4, breaks=breaks, save_t=ts, obs=obs, fused=fused)
    assert stats["n_failed"] == 0
    for i, pt in enumerate(points):
        params = {"vdd": pt.get("vdd", 5.0)}
        # the port starts from the GPU's DC state: the flop's DC point is not unique (see test_gpu_drivers)
        ref, rst = _port_run(circ, params, pt.get("temp", 27.0), u0[i], ts, obs, sim.vscale())
        assert rst["status"] == 1
        # identical decision path: same Newton / step / reject counts.  (The fused kernel accumulates stamps in a
        # different order -- rounding-level differences in the residual -- so its counts are only required to be close.)
        if fused == 0:
            assert (per[i, 0], per[i, 1], per[i, 2]) == (rst["newton_iters"], rst["accepted"], rst["rejected"]), (pt, per[i], rst)
        else:
            assert abs(per[i, 0] - rst["newton_iters"]) <= 0.01 * rst["newton_iters"], (pt, per[i], rst)
        err = np.max(np.abs(out[i] - ref) / np.maximum(np.abs(ref), 1.0))
        assert err <= REL_TOL, (pt, err)
    sim.close()
