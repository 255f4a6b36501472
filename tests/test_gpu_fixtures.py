"""GPU mirrors of the reference's numeric sp_mos1 fixtures (tests/test_oracle_golden.py holds the oracle's side):
the HIP path's DC solution and its G / C stamps, read back through the C ABI (cadnip_dc_run, cadnip_rebuild, cadnip_get_GCb),
must reproduce test/opinfo.jl:158-205 (level-1 gm, gds, vgs, vds) and the ngspice-43 AC table of test/ac.jl:204-272."""
import numpy as np
import pytest
import scipy.sparse as sp

from cadnip_jl_amd import api
from oracle import mna_ref as M
from tests import circuits as tc
from tests.test_oracle_golden import check_against_ngspice, load_ngspice_inverter

pytestmark = pytest.mark.gpu


def _gpu_dc_system(circ, fused=False):
    sim = api.BatchSimulator(api.MNACircuit(circ, {}, api.MNASpec(mode="dcop")))
    try:
        u, conv, _ = sim.dc(fused=fused)
        assert conv[0]
        sim.h.rebuild(u, 0.0)
        G, C, b, _ = sim.h.get_GCb()
        st = sim.st
        dense = lambda nz: sp.csc_matrix((nz, st.ref_rowval, st.ref_colptr), shape=(st.n, st.n)).toarray()
        return st, u[0], dense(G[0]), dense(C[0])
    finally:
        sim.close()


@pytest.mark.parametrize("fused", [False, True])
def test_gpu_mos1_cs_stage_small_signal_numbers(fused):
    VOV = 1.1472 - 0.7
    GM = 100e-6 * 20.0 * VOV
    st, x, G, _ = _gpu_dc_system(tc.cs_stage(), fused)
    d, g = st.index_of("drain"), st.index_of("gate")
    i_d = -x[st.index_of("I_vdd")]
    assert i_d == pytest.approx(200e-6, rel=0.05)
    assert i_d == pytest.approx((5.0 - x[d]) / 10e3, rel=1e-6)
    gm = G[d, g]
    assert gm == pytest.approx(GM, rel=0.05)
    cur = []
    for vb in (1.1472 + 1e-3, 1.1472 - 1e-3):
        st2, x2, _, _ = _gpu_dc_system(tc.cs_stage(vb), fused)
        cur.append(-x2[st2.index_of("I_vdd")])
    assert gm == pytest.approx((cur[0] - cur[1]) / 2e-3, rel=0.02)
    assert G[d, d] - 1e-4 == pytest.approx(0.01 * i_d, rel=0.10)
    assert x[st.index_of("m1_sp_mos1_lim_g_s_int")] == pytest.approx(1.1472, rel=1e-6)
    assert x[st.index_of("m1_sp_mos1_lim_d_int_s_int")] == pytest.approx(x[d], rel=1e-6)
    assert x[d] > VOV


@pytest.mark.parametrize("fused", [False, True])
def test_gpu_mos1_inverter_ac_matches_ngspice_table(fused):
    freqs, ref = load_ngspice_inverter()
    st, x, G, C = _gpu_dc_system(tc.cmos_inverter_ac(), fused)
    b_ac = np.zeros(st.n)
    b_ac[st.index_of("I_vin")] = 1.0
    resp = M.ac_response(G, C, b_ac, 2 * np.pi * freqs, st.n_nodes)[:, st.index_of("vout")]
    check_against_ngspice(resp, ref)
    assert np.allclose(np.abs(resp), np.abs(ref), rtol=1e-4, atol=0.0) and np.max(np.abs(np.angle(resp) - np.angle(ref))) < 1e-4
