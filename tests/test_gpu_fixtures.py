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


# ---- test/opinfo.jl: device terminal currents and operating-point variables from the DC solution -------------------------------
def test_gpu_opinfo_terminal_currents():
    import cadnip_jl_amd as cj
    # resistors (opinfo.jl:55-71): divider 6 V, 1k / 2k
    c = cj.Circuit("divider")
    c.V("v1", "in", "0", dc=6.0)
    c.R("r1", "in", "out", 1e3)
    c.R("r2", "out", "0", 2e3)
    op = api.dc(api.MNACircuit(c, {}))
    I = 6.0 / 3000.0
    assert op["i_r1_p"] == pytest.approx(I) and op["i_r1_n"] == pytest.approx(-I) and op["i_r2_p"] == pytest.approx(I) and op["i_r2_n"] == pytest.approx(-I)
    assert abs(op["i_r1_n"] + op["i_r2_p"]) <= 1e-15 and abs(op["I_v1"]) == pytest.approx(I)
    assert set(op.terminal_currents()) >= {"i_r1_p", "i_r1_n", "i_r2_p", "i_r2_n", "i_v1_p", "i_v1_n"} and "i_r9_p" not in op
    # diode (opinfo.jl:73-83): the junction current is the loop current
    c = cj.Circuit("rectifier")
    c.V("v1", "in", "0", dc=5.0)
    c.R("r1", "in", "out", 1e3)
    c.D("d1", "out", "0", Is=76.9e-12, n_=1.45)
    op = api.dc(api.MNACircuit(c, {}))
    assert 0.6 < op["out"] < 0.8
    ir = (op["in"] - op["out"]) / 1e3
    assert op["i_d1_a"] == pytest.approx(ir, rel=1e-6) and op["i_d1_c"] == pytest.approx(-ir, rel=1e-6) and op["i_r1_n"] == pytest.approx(-ir, rel=1e-9)
    assert op["d1_gd"] == pytest.approx(op["i_d1_a"] / (1.45 * 0.026), rel=0.05)      # opinfo.jl:207-213 (this Diode takes Vt = 26 mV)
    assert op["d1_vd"] == pytest.approx(op["out"], rel=1e-6)


def test_gpu_opinfo_mos1_operating_point_variables():
    """test/opinfo.jl:85-147, 158-205 on the level-1 common-source stage: drain current without inferring it, KCL over the four
    terminals, and the model's own small-signal numbers (gm = K VOV, gds = lambda ID, vdsat = VOV)."""
    VOV = 1.1472 - 0.7
    GM = 100e-6 * 20.0 * VOV
    op = api.dc(api.MNACircuit(tc.cs_stage(), {}))
    assert op["i_m1_d"] == pytest.approx(200e-6, rel=0.05)
    assert op["i_m1_d"] == pytest.approx(-op["I_vdd"], rel=1e-6) and op["i_m1_d"] == pytest.approx(op["i_rd_p"], rel=1e-6)
    assert abs(op["i_m1_g"]) <= 1e-9 and abs(op["i_m1_d"] + op["i_m1_g"] + op["i_m1_s"] + op["i_m1_b"]) <= 1e-9
    assert op["m1_gm"] == pytest.approx(GM, rel=0.05)
    up = api.dc(api.MNACircuit(tc.cs_stage(1.1472 + 1e-3), {}))
    dn = api.dc(api.MNACircuit(tc.cs_stage(1.1472 - 1e-3), {}))
    assert op["m1_gm"] == pytest.approx((up["i_m1_d"] - dn["i_m1_d"]) / 2e-3, rel=0.02)
    assert op["m1_gds"] == pytest.approx(0.01 * op["i_m1_d"], rel=0.10)
    assert op["m1_vdsat"] == pytest.approx(VOV, rel=0.05) and op["m1_vds"] > op["m1_vdsat"]
    assert op["m1_vgs"] == pytest.approx(1.1472, rel=1e-6) and op["m1_vds"] == pytest.approx(op["drain"], rel=1e-6)
    hot = api.dc(api.MNACircuit(tc.cs_stage(1.20), {}))
    assert hot["i_m1_d"] > op["i_m1_d"] * 1.1                                   # opinfo.jl:139-146
    assert set(op.op_vars()) >= {"m1_gm", "m1_gds", "m1_vdsat", "m1_vgs", "m1_vds", "m1_von", "m1_gmbs"}
