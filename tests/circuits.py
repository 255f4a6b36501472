"""Test circuits shared by the CPU (oracle) and GPU (parity) suites."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cadnip_jl_amd as cj
from cadnip_jl_amd import benchmarks as bm

NM = dict(bm.NFET_06V0_MEYER)
PM = dict(bm.PFET_06V0_MEYER)


def divider(v=5.0, r1=1e3, r2=1e3):
    c = cj.Circuit()
    c.V("v1", "vcc", "0", dc=v)
    c.R("r1", "vcc", "out", r1)
    c.R("r2", "out", "0", r2)
    return c


def linear_zoo():
    """Every linear builtin once (R C L V I VCVS VCCS CCVS CCCS) with PWL / PULSE / SIN sources."""
    c = cj.Circuit()
    c.V("v1", "a", "0", dc=1.5, wave=("pwl", [0.0, 1e-3, 2e-3, 2e-3, 5e-3], [0.0, 1.0, 1.0, 3.0, -1.0]))
    c.V("v2", "b", "0", dc=0.3, wave=("pulse", 0.0, 2.0, 1e-4, 2e-4, 3e-4, 5e-4, 2e-3))
    c.I("i1", "c", "0", dc=1e-3, wave=("sin", 0.1e-3, 1e-3, 1e3, 2e-4, 50.0, 30.0))
    c.R("r1", "a", "c", 1e3)
    c.R("r2", "b", "c", 2.2e3)
    c.C("c1", "c", "0", 1e-6)
    c.L("l1", "c", "d", 1e-3)
    c.R("r3", "d", "0", 50.0)
    c.E("e1", "e", "0", "c", "d", 2.0)
    c.R("r4", "e", "f", 1e3)
    c.G("g1", "f", "0", "a", "b", 1e-3)
    c.R("r5", "f", "0", 3e3)
    c.H("h1", "g", "0", "f", "h", 100.0)
    c.R("r6", "h", "0", 1e3)
    c.R("r7", "g", "0", 1e3)
    c.F("f1", "k", "0", "g", "m", 3.0)
    c.R("r8", "m", "0", 500.0)
    c.R("r9", "k", "0", 1e3)
    return c


def diode_rectifier(limit=True):
    c = cj.Circuit()
    c.V("v1", "vin", "0", dc=5.0)
    c.R("r1", "vin", "out", 1e3)
    c.D("d1", "out", "0", Is=1e-14, limit=limit)
    return c


def diode_chain():
    """3-diode 50 V chain of test/mna/pcnr.jl:330-350."""
    c = cj.Circuit()
    c.V("v1", "n0", "0", dc=50.0)
    c.R("r1", "n0", "n1", 1e3)
    c.D("d1", "n1", "n2", Is=1e-14)
    c.D("d2", "n2", "n3", Is=1e-14)
    c.D("d3", "n3", "0", Is=1e-14)
    return c


def nonlinear_zoo():
    c = cj.Circuit()
    c.V("v1", "vdd", "0", dc=3.0)
    c.V("v2", "in", "0", dc=1.2, wave=("pwl", [0.0, 1e-6], [0.0, 3.0]))
    c.R("r1", "vdd", "out", 10e3)
    c.SMOS("m1", "out", "in", "0", Vth=0.5, K=1e-3, lambda_=0.02)
    c.DCAP("d1", "out", "x", Is=1e-14, Cj0=2e-12)
    c.R("r2", "x", "0", 1e4)
    c.D("d2", "in", "y", limit=False)
    c.R("r3", "y", "0", 1e5)
    return c


def rc_charge(v=5.0, r=1e3, c_=1e-6):
    c = cj.Circuit()
    c.V("v1", "vin", "0", dc=v)
    c.R("r1", "vin", "out", r)
    c.C("c1", "out", "0", c_)
    return c


def inverter_dc(vin=2.5):
    c = cj.Circuit()
    c.V("vdd", "vdd", "0", dc=5.0)
    c.V("vin", "in", "0", dc=vin)
    c.MOS1("mn", "out", "in", "0", "0", NM, l=0.6e-6, w=0.36e-6)
    c.MOS1("mp", "out", "in", "vdd", "vdd", PM, l=0.5e-6, w=0.495e-6)
    c.C("cl", "out", "0", 1e-15)
    return c


def mos1_rd():
    """sp_mos1 with rd/rs given -> genuine internal nodes d_int, s_int; nsub/tox-derived card."""
    c = cj.Circuit()
    card = dict(type=1, tox=2e-8, nsub=1e16, u0=500.0, rd=20.0, rs=15.0, cj=2e-4, cjsw=1e-10, mj=0.4, mjsw=0.3, js=1e-6,
                cgso=2e-10, cgdo=2e-10, cgbo=1e-10, ld=5e-8)
    card["lambda"] = 0.05
    c.V("vd", "d", "0", dc=2.0)
    c.V("vg", "g", "0", dc=1.5)
    c.R("rs", "s", "0", 100.0)
    c.V("vb", "b", "0", dc=-0.5)
    c.MOS1("m1", "d", "g", "s", "b", card, l=1e-6, w=10e-6, ad=1e-11, pd=2e-5, ps=2e-5, m=2.0, **{"as": 1e-11})
    return c


def behavioral():
    """Behavioural V and I sources (devices.jl:1003-1131) over node voltages and time; the current source is a
    contraction (|d i/d v| R < 1) so the fixed-source Newton iteration of the reference converges."""
    c = cj.Circuit()
    c.V("v1", "in", "0", dc=2.0)
    c.R("r1", "in", "x", 1.0)
    c.BI("b1", "x", "0", "-0.1*V(x)**2")
    c.BV("b2", "y", "0", "2*V(x) + tanh(V(in, x)) - min(V(x), 0.5)*exp(-abs(V(in))) + 1e3*t", scale=0.5)
    c.R("r2", "y", "z", 1e3)
    c.BI("b3", "z", "0", "sqrt(abs(V(y))) * cos(V(x)) / max(1, V(in)^2)")
    c.R("r3", "z", "0", 2e3)
    return c


def ring_oscillator():
    """3-stage CMOS ring oscillator of the reference's sp_mos1 test (test/mna/vadistiller_integration.jl:45-60):
    level-1 cards vto = -/+0.7, kp = 50u / 100u, W/L = 2u/1u and 1u/1u, Vdd = 3.3 V, 10 fF on every stage output."""
    c = cj.Circuit("3-stage CMOS ring oscillator")
    pm, nm = dict(type=-1, vto=-0.7, kp=50e-6), dict(type=1, vto=0.7, kp=100e-6)
    c.V("Vdd", "vdd", "0", dc=3.3)
    for k, (o, i) in enumerate((("out1", "in1"), ("out2", "out1"), ("in1", "out2")), 1):
        c.MOS1("MP%d" % k, o, i, "vdd", "vdd", pm, w=2e-6, l=1e-6)
        c.MOS1("MN%d" % k, o, i, "0", "0", nm, w=1e-6, l=1e-6)
    c.C("C1", "out1", "0", 10e-15)
    c.C("C2", "out2", "0", 10e-15)
    c.C("C3", "in1", "0", 10e-15)
    return c


def va_zoo():
    """Every Verilog-A module that is compiled into the library (cadnip.jl_amd/va/models): a resistor / diode (internal
    node, voltage-dependent charge) / capacitor (constant capacitance) chain and a CMOS inverter from va_sqmos."""
    c = cj.Circuit("generated Verilog-A modules")
    c.V("Vin", "in", "0", dc=0.9, wave=("sin", 0.9, 0.4, 2e6, 0.0, 0.0, 0.0))
    c.V("Vdd", "vdd", "0", dc=1.8)
    c.VA("XR1", "va_resistor", ("in", "a"), r=2e3)
    c.VA("XD1", "va_diode", ("a", "k"), rs=25.0, cj0=2e-12, tt=5e-9)
    c.VA("XR2", "va_resistor", ("k", "0"), r=1e3, m=2.0)
    c.VA("XC1", "va_capacitor", ("k", "0"), c=3e-12)
    c.VA("XD2", "va_diode", ("0", "k"), **{"is": 5e-14, "rs": 0.0})       # rs = 0: V(a,ai) <+ 0 collapses the internal node
    c.VA("XMN", "va_sqmos", ("out", "in", "0", "0"), type=1.0, w=2e-6)
    c.VA("XMP", "va_sqmos", ("out", "in", "vdd", "vdd"), type=-1.0, vto=0.7, kp=40e-6, w=4e-6)
    c.VA("XCL", "va_capacitor", ("out", "0"), c=20e-15)
    return c


def va_limited():
    """$limit sites: the generated limited diode (va_dlim: user pnjlim through $limit, limit unknown, lim_rhs terms) as a
    half-wave rectifier and a two-diode clamp.  (No built-in limited diode beside it: the reference's own `limit!` reads
    x[li] unguarded and would throw during the detection passes once a junction charge shifts the limit indices,
    devices.jl:1213-1217.)"""
    c = cj.Circuit("generated Verilog-A module with $limit")
    c.V("Vs", "in", "0", dc=2.0, wave=("sin", 0.3, 2.0, 1e6, 0.0, 0.0, 0.0))
    c.VA("XD1", "va_dlim", ("in", "out"), cj=2e-12)
    c.R("RL", "out", "0", 1e3)
    c.C("CL", "out", "0", 1e-9)
    c.VA("XR", "va_resistor", ("in", "x"), r=500.0)
    c.VA("XD2", "va_dlim", ("x", "0"), **{"is": 2e-14, "n": 1.5})
    c.VA("XD3", "va_dlim", ("0", "x"))
    return c


VA_NCARD = dict(type=1.0, vto=0.7, kp=100e-6, gamma=0.45, phi=0.7, cbd=1e-15, cbs=1e-15, pb=0.8, **{"lambda": 0.03})
VA_PCARD = dict(type=-1.0, vto=-0.8, kp=50e-6, gamma=0.45, phi=0.7, cbd=1e-15, cbs=1e-15, pb=0.8, **{"lambda": 0.04})


def va_mos_inverter(builtin=False, rd=0.0):
    """CMOS inverter + pass transistor from the generated level-1 MOSFET (va_mos1l: three $limit sites, two junction
    charges, series resistances that collapse when zero) -- or, with ``builtin``, the same netlist from the hand-written
    sp_mos1 device with the same card (benchmarks.py NFET_06V0 / PFET_06V0)."""
    c = cj.Circuit("level-1 MOSFETs: generated vs hand-written")
    c.V("Vdd", "vdd", "0", dc=5.0)
    c.V("Vin", "in", "0", dc=1.2, wave=("pwl", [0.0, 2e-9, 4e-9, 12e-9, 14e-9], [1.2, 1.2, 4.0, 4.0, 0.5]))
    fets = (("mn", ("out", "in", "0", "0"), "n", 3.6e-7, 6e-7), ("mp", ("out", "in", "vdd", "vdd"), "p", 4.95e-7, 5e-7),
            ("mt", ("y", "vdd", "out", "0"), "n", 3.6e-7, 6e-7))
    for name, nodes, pol, w, l in fets:
        if builtin:
            card = dict(bm.NFET_06V0 if pol == "n" else bm.PFET_06V0)
            if rd:
                card.update(rd=rd, rs=rd)
            c.MOS1(name, nodes[0], nodes[1], nodes[2], nodes[3], card, w=w, l=l)
        else:
            c.VA(name, "va_mos1l", nodes, w=w, l=l, rd=rd, rs=rd, **(VA_NCARD if pol == "n" else VA_PCARD))
    c.C("cl", "out", "0", 5e-15)
    c.C("cy", "y", "0", 2e-15)
    c.R("ry", "y", "0", 1e6)
    return c


def ring_checks(v):
    """The reference's assertions on V(out1) sampled at 500 points over the last 100 ns (vadistiller_integration.jl:668-690)."""
    import numpy as np
    lo, hi = float(np.min(v)), float(np.max(v))
    mid = 0.5 * (lo + hi)
    crossings = int(np.sum(((v[:-1] < mid) & (v[1:] >= mid)) | ((v[:-1] > mid) & (v[1:] <= mid))))
    assert hi - lo > 2.0 and hi > 2.5 and lo < 0.8, (lo, hi)
    assert crossings > 10, crossings
    return lo, hi, crossings


def va_feat():
    """Two instances of va_feat (cadnip.jl_amd/va/models): one with tc given (so $param_given differs between them), level 1 and
    the internal node collapsed; one at level 3 with a series resistance (genuine internal node), through the alias gg."""
    c = cj.Circuit("va_feat")
    c.V("v1", "in", "0", dc=1.0)
    c.R("r1", "in", "p", 50.0)
    c.VA("x1", "va_feat", ("p", "q", "r"), g0=2e-3, tc=1e-3, cj=2e-12)
    c.VA("x2", "va_feat", ("q", "0", "r"), gg=5e-4, rs=5.0, lev=3)
    c.R("r2", "r", "0", 1e3)
    c.R("r3", "q", "0", 2e3)
    return c


ALL_STAMP = {
    "divider": (divider, {}), "linear_zoo": (linear_zoo, {}), "diode": (diode_rectifier, {}),
    "diode_nolimit": (lambda: diode_rectifier(False), {}), "diode_chain": (diode_chain, {}),
    "nonlinear_zoo": (nonlinear_zoo, {}), "behavioral": (behavioral, {}), "inverter": (inverter_dc, {}), "mos1_rd": (mos1_rd, {}),
    "va_zoo": (va_zoo, {}), "va_feat": (va_feat, {}), "va_limited": (va_limited, {}), "va_mos_inverter": (va_mos_inverter, {}),
    "va_mos_inverter_rd": (lambda: va_mos_inverter(rd=40.0), {}),
    "dff": (bm.dff_circuit, {"vdd": 5.0}), "dff_meyer": (lambda: bm.dff_circuit(meyer=True), {"vdd": 5.0}),
}


def cs_stage(vbias=1.1472):
    """Common-source stage of the reference's operating-point tests (test/opinfo.jl:35-42): level-1 NMOS vto = 0.7,
    kp = 100u, lambda = 0.01, W/L = 20u/1u, Rd = 10k to a 5 V supply, gate at vbias (design bias 1.1472 V: VOV = 447.2 mV)."""
    c = cj.Circuit("cs_stage")
    card = dict(type=1, vto=0.7, kp=100e-6)
    card["lambda"] = 0.01
    c.V("vdd", "vdd", "0", dc=5.0)
    c.V("vin", "gate", "0", dc=vbias)
    c.MOS1("m1", "drain", "gate", "0", "0", card, w=20e-6, l=1e-6)
    c.R("rd", "vdd", "drain", 10e3)
    return c


def cmos_inverter_ac():
    """CMOS inverter of the reference's AC test against ngspice-43 (test/ac.jl:246-256): level-1 cards vto = -/+0.7,
    kp = 50u / 100u, lambda = 0.01, cgso = cgdo = 1e-15 F/m, W/L = 2u/1u and 1u/1u, Cload = 10 fF, Vdd = 3.3 V,
    Vin = 1.65 V DC (+ 1 V AC: the excitation is applied by the test on Vin's branch row)."""
    c = cj.Circuit("cmos_inverter_ac")
    pm = dict(type=-1, vto=-0.7, kp=50e-6, cgso=1e-15, cgdo=1e-15)
    nm = dict(type=1, vto=0.7, kp=100e-6, cgso=1e-15, cgdo=1e-15)
    pm["lambda"] = nm["lambda"] = 0.01
    c.V("vdd", "vdd", "0", dc=3.3)
    c.V("vin", "vin", "0", dc=1.65)
    c.MOS1("mp", "vout", "vin", "vdd", "vdd", pm, w=2e-6, l=1e-6)
    c.MOS1("mn", "vout", "vin", "0", "0", nm, w=1e-6, l=1e-6)
    c.C("cload", "vout", "0", 10e-15)
    return c


# ---- circuits that walk the DC fallback chain (solve.jl:871-929) past its first stage -----------------------------------
def bi_quadratic(k=0.05):
    """2 V through 1 ohm into a node loaded by 1 ohm and a behavioural current sink k V^2 (a fixed-source iteration,
    devices.jl:1079-1131): with abstol 1e-7 and maxiters 5 plain Newton and every gshunt rung run out of iterations from a
    cold start, the warm-started source ramp arrives (found by scanning k / abstol / maxiters on the oracle)."""
    c = cj.Circuit("bi_quadratic")
    c.V("v1", "in", "0", dc=2.0)
    c.R("r1", "in", "x", 1.0)
    c.BI("b1", "x", "0", "%g*V(x)**2" % k)
    c.R("r2", "x", "0", 1.0)
    return c


def inverter_chain(stages=8, vin=0.0):
    """Chain of level-1 CMOS inverters (the benchmark cards): PCNR needs 16 iterations at 8 stages; plain Newton and the
    homotopies meet non-finite stamps (the model has no sub-threshold current: DESIGN.md section 5)."""
    c = cj.Circuit("inverter_chain")
    c.V("vdd", "vdd", "0", dc=5.0)
    c.V("vin", "n0", "0", dc=vin)
    for k in range(stages):
        c.MOS1("mn%d" % k, "n%d" % (k + 1), "n%d" % k, "0", "0", dict(bm.NFET_06V0), l=0.6e-6, w=0.36e-6)
        c.MOS1("mp%d" % k, "n%d" % (k + 1), "n%d" % k, "vdd", "vdd", dict(bm.PFET_06V0), l=0.5e-6, w=0.495e-6)
    return c
