"""ac! on the GPU path (src/ac.jl:113-170): DC operating point and linearisation on the device, the
reference's dense frequency sweep on the host.  Fixtures: test/ac.jl -- the third-order Butterworth low-pass against its
transfer function (:18-94), source phase (:101-108), an AC current source (:142-148), and the ngspice-43 table of the sp_mos1
CMOS inverter (:204-272) through the product API."""
import numpy as np
import pytest

import cadnip_jl_amd as cj
from cadnip_jl_amd import api, netlist
from tests import circuits as tc
from tests.test_oracle_golden import check_against_ngspice, load_ngspice_inverter

pytestmark = pytest.mark.gpu

BUTTERWORTH = """*Third order low pass filter, butterworth, with w_c = 1
.param res=1
V1 vin 0 AC 1
L1 vin n1 1.5
C2 n1 0 1.3333333333333333
L3 n1 vout 0.5
R4 vout 0 '2*res'
R5 vout 0 '2*res'
"""


def test_butterworth_low_pass_against_its_transfer_function():
    circ, _ = netlist.read_spice(BUTTERWORTH)
    freqs = api.acdec(20, 0.01, 10)
    assert len(freqs) == 61 and freqs[0] == 0.01 and freqs[-1] == pytest.approx(10.0)
    sol = api.ac(api.MNACircuit(circ, {}), freqs)
    w = 2 * np.pi * freqs
    s = 1j * w
    H = 1.0 / ((s + 1.0) * (s * s + s + 1.0))
    resp = sol.freqresp("vout", w)
    assert np.allclose(resp, H, rtol=1e-9, atol=0.0)                      # ac.jl:50
    assert np.allclose(sol.freqresp("vin", w), 1.0)                      # ac.jl:52: the directly observed source
    assert np.array_equal(sol["vout"], resp) and np.allclose(sol["vin"], 1.0)                               # ac.jl:56-57
    assert np.allclose(sol.magnitude_db("vout"), 20 * np.log10(np.abs(H))) and np.allclose(sol.phase_deg("vout"), np.degrees(np.angle(H)))
    assert np.allclose(sol.magnitude_db("vout", freqs), 20 * np.log10(np.abs(H)))                            # ac.jl:64
    nogrid = api.ac(api.MNACircuit(circ, {}))
    assert len(nogrid["vout"]) == 0 and np.allclose(nogrid.freqresp("vout", w), resp)                       # ac.jl:70-72
    VL3 = sol.freqresp("n1", w) - sol.freqresp("vout", w)
    assert np.allclose(VL3, s * 0.5 * H, rtol=1e-9)                        # ac.jl:88-94: V = s L3 H(s)


def test_source_phase_and_current_source_excitation():
    circ, _ = netlist.read_spice("* AC source with explicit phase\nV1 vin 0 AC 1 90\nR1 vin 0 1k\n")
    sol = api.ac(api.MNACircuit(circ, {}))
    assert np.allclose(sol.freqresp("vin", [1.0, 10.0]), 1.0j)             # ac.jl:101-108
    c = cj.Circuit("isource ac")
    c.I("i1", "vin", "0", ac=1.0)                                           # the AC current enters p (devices.jl:693-729): 1 A into 1 Ohm (ac.jl:142-148)
    c.R("r1", "vin", "0", 1.0)
    sol = api.ac(api.MNACircuit(c, {}))
    assert np.allclose(sol.freqresp("vin", [1.0, 10.0]), 1.0 + 0.0j, rtol=1e-8)


def test_mos1_inverter_ac_through_the_product_api():
    freqs, ref = load_ngspice_inverter()
    circ = tc.cmos_inverter_ac()
    next(d for d in circ.devices if d.name == "vin").params["ac"] = 1.0
    sol = api.ac(api.MNACircuit(circ, {}), freqs)
    check_against_ngspice(sol["vout"], ref)                                 # test/ac.jl:267-272
    assert np.allclose(np.abs(sol["vout"]), np.abs(ref), rtol=1e-4, atol=0.0)
    # a sweep of the supply: one ACSol per point from one resident batch; the gain peaks where both devices saturate
    cs = api.CircuitSweep(api.MNACircuit(_with_param_vdd(), {"vdd": 3.3}), api.Sweep(vdd=[3.0, 3.3, 3.6]))
    res = api.ac(cs, freqs[:3])
    g = [abs(res[i]["vout"][0]) for i in range(3)]
    assert g[1] == pytest.approx(abs(ref[0]), rel=1e-4) and len(set(np.round(g, 6))) == 3


def _with_param_vdd():
    c = tc.cmos_inverter_ac()
    for d in c.devices:
        if d.name == "vdd":
            d.params["dc"] = cj.Param("vdd")
        if d.name == "vin":
            d.params["dc"] = cj.Param("vdd", scale=0.5)
            d.params["ac"] = 1.0
    return c
