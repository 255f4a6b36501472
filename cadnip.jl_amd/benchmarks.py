"""Synthetic benchmark circuits named by BASELINE.json.

Topology and stimulus are the reference's gf180 D-flip-flop deck
(/root/reference/test/DFF/DFF_cap_all.cir and
/root/reference/test/DFF/gf180mcu_fd_sc_mcu7t5v0__dffnq_4.ngspice: 30 MOSFETs, 7 V-sources,
CQ = 172.05 fF, PWL clock / data).  The gf180 PDK model cards (sm141064.ngspice) are not in the
reference repository, so -- as BASELINE.json says ("synthetic gf180 DFF") -- the devices use the
in-repo level-1 model sp_mos1 (models/VADistillerModels.jl/va/mos1.va) with the documented
synthetic card below.  V-sources come first, as the reference's SPICE codegen orders them
(src/spc/codegen.jl:3130-3149); MOSFET terminals are d g s b as in the deck.
"""
import numpy as np

from .circuit import Circuit, Param

# Synthetic 5 V CMOS level-1 cards (documented, not a PDK extraction).
#
# Two variants:
#  * *_MEYER: tox given -> sp_mos1's Meyer gate-charge model is active.  Cadnip lowers
#    ``capgs*ddt(vgs)`` through its s-dual into a branch *charge* q = capgs(V)*vgs
#    (contrib.jl:356-375), whose dq/dV jumps by ~10x at the Meyer region boundaries.  At hot
#    corners the resulting DAE has impasse points (no solution of the step equation across the
#    kink, for any step size -- tools/dbg_root.py), so this variant is used for stamp / DC /
#    nominal-corner parity tests only.
#  * default (benchmark) cards: no tox -> OxideCap = 0, the Meyer charge is identically zero
#    (meyer_scale = 0, mos1.va:1042-1048, which also removes the cgso/cgdo overlap terms); the
#    gate capacitance is lumped into fixed Cgs / Cgd capacitors per instance
#    (0.5*Cox*W*L + Cov*W each, Cox = 3.9 eps0 / 15 nm, Cov = 0.3 nF/m).  Junction charges
#    (cbd = cbs = 1 fF, depletion law mos1.va:1049-1109) stay voltage dependent.
NFET_06V0 = dict(type=1, vto=0.7, kp=100e-6, gamma=0.45, phi=0.7, cbd=1e-15, cbs=1e-15, pb=0.8)
NFET_06V0["lambda"] = 0.03
PFET_06V0 = dict(type=-1, vto=-0.8, kp=50e-6, gamma=0.45, phi=0.7, cbd=1e-15, cbs=1e-15, pb=0.8)
PFET_06V0["lambda"] = 0.04
NFET_06V0_MEYER = dict(NFET_06V0, tox=1.5e-8, cgso=0.3e-9, cgdo=0.3e-9)
PFET_06V0_MEYER = dict(PFET_06V0, tox=1.5e-8, cgso=0.3e-9, cgdo=0.3e-9)
COX_AREA = 3.9 * 8.854214871e-12 / 1.5e-8   # F/m^2
COV_WIDTH = 0.3e-9                          # F/m


def lumped_gate_cap(W, L):
    return 0.5 * COX_AREA * W * L + COV_WIDTH * W

_P = 1e-12
CLKN_PWL = ([0.0, 50000 * _P, 51020 * _P, 100000 * _P, 101020 * _P, 400000 * _P, 401020 * _P, 500000 * _P, 501020 * _P,
             600000 * _P, 601020 * _P, 700000 * _P],
            [5.0, 5.0, 0.0, 0.0, 5.0, 5.0, 0.0, 0.0, 5.0, 5.0, 0.0, 0.0])
D_PWL = ([0.0, 200000 * _P, 201020 * _P, 300000 * _P, 301020 * _P, 400000 * _P, 401020 * _P, 600000 * _P],
         [0.0, 0.0, 5.0, 5.0, 0.0, 0.0, 5.0, 5.0])

# (name, d, g, s, b, W, L) -- the .ngspice subcircuit, instance order preserved
DFF_FETS = [
    ("tn10", "VSS", "D", "D_neg", "VPW", 3.6e-07, 6e-07), ("tp10", "VDD", "D", "D_neg", "VNW", 4.95e-07, 5e-07),
    ("tn11", "D_neg", "cki", "D_neg_clked", "VPW", 3.6e-07, 6e-07), ("tp11", "D_neg_clked", "ncki", "D_neg", "VNW", 4.95e-07, 5e-07),
    ("tn15", "Q_internal", "D_neg_clked", "VSS", "VPW", 3.6e-07, 6e-07), ("tp15", "Q_internal", "D_neg_clked", "VDD", "VNW", 4.95e-07, 5e-07),
    ("tn0", "D_neg_clked", "ncki", "net11", "VPW", 3.6e-07, 6e-07), ("tp0", "net4", "cki", "D_neg_clked", "VNW", 4.95e-07, 5e-07),
    ("tn1", "VSS", "Q_internal", "net11", "VPW", 3.6e-07, 6e-07), ("tp1", "VDD", "Q_internal", "net4", "VNW", 4.95e-07, 5e-07),
    ("tn2", "net0", "ncki", "Q_internal", "VPW", 3.6e-07, 6e-07), ("tp7", "net0", "cki", "Q_internal", "VNW", 4.95e-07, 5e-07),
    ("tn3", "net7", "cki", "net0", "VPW", 3.6e-07, 6e-07), ("tp6", "net7", "ncki", "net0", "VNW", 4.95e-07, 5e-07),
    ("tn5", "Q_neg", "net0", "VSS", "VPW", 9.45e-07, 6e-07), ("tp3", "Q_neg", "net0", "VDD", "VNW", 1.075e-06, 5e-07),
    ("tn4", "VSS", "Q_neg", "net7", "VPW", 9.45e-07, 6e-07), ("tp2", "VDD", "Q_neg", "net7", "VNW", 1.075e-06, 5e-07),
    ("tn6_7", "Q", "Q_neg", "VSS", "VPW", 8.2e-07, 6e-07), ("tn6", "Q", "Q_neg", "VSS", "VPW", 8.2e-07, 6e-07),
    ("tn6_7_61", "Q", "Q_neg", "VSS", "VPW", 8.2e-07, 6e-07), ("tn6_49", "Q", "Q_neg", "VSS", "VPW", 8.2e-07, 6e-07),
    ("tp4_13", "Q", "Q_neg", "VDD", "VNW", 10.95e-07, 5e-07), ("tp4", "Q", "Q_neg", "VDD", "VNW", 10.95e-07, 5e-07),
    ("tp4_13_64", "Q", "Q_neg", "VDD", "VNW", 10.95e-07, 5e-07), ("tp4_55", "Q", "Q_neg", "VDD", "VNW", 10.95e-07, 5e-07),
    ("tn9", "ncki", "CLKN", "VSS", "VPW", 4.65e-07, 6e-07), ("tp9", "ncki", "CLKN", "VDD", "VNW", 8.65e-07, 5e-07),
    ("tn16", "cki", "ncki", "VSS", "VPW", 4.65e-07, 6e-07), ("tp16", "cki", "ncki", "VDD", "VNW", 8.65e-07, 5e-07),
]


def _card(base, **over):
    c = dict(base)
    c.update(over)
    return c


def dff_circuit(mc_vto=None, mc_kp=None, meyer=False, generated=False):
    """gf180 DFF (dffnq_4) test bench.  ``generated``: the MOSFETs are instances of the generated Verilog-A level-1
    module ``va_mos1l`` (cadnip.jl_amd/va/models) instead of the hand-written sp_mos1 device -- same cards, same netlist.  Sweepable parameter ``vdd`` (default 5 V) scales the
    supply and the PWL stimulus amplitude together.  ``mc_vto`` / ``mc_kp`` (optional Param names)
    add Monte-Carlo shifts: vto += type * params[mc_vto], kp *= params[mc_kp].  ``meyer`` selects
    the Meyer-charge card variant (see the card comment above)."""
    c = Circuit("gf180 dffnq_4 test bench (synthetic level-1 cards)")
    amp = Param("vdd", scale=1.0 / 5.0)
    c.V("VVDD", "VDD", "0", dc=Param("vdd"))
    c.V("VVSS", "VSS", "0", dc=0.0)
    c.V("VQ", "Q", "Q_tmp", dc=0.0)
    c.V("VNW", "VNW", "VDD", dc=0.0)
    c.V("VPW", "VPW", "VSS", dc=0.0)
    # DC value of a PWL source = its first point (src/spc/codegen.jl:2597), here scaled with the supply like the wave
    c.V("VCLKN", "CLKN", "0", dc=Param("vdd", scale=CLKN_PWL[1][0] / 5.0), wave=("pwl",) + CLKN_PWL, scale=amp)
    c.V("VD", "D", "0", dc=Param("vdd", scale=D_PWL[1][0] / 5.0), wave=("pwl",) + D_PWL, scale=amp)
    for (nm, d, g, s, b, W, L) in DFF_FETS:
        if meyer:
            base = NFET_06V0_MEYER if nm.startswith("tn") else PFET_06V0_MEYER
        else:
            base = NFET_06V0 if nm.startswith("tn") else PFET_06V0
        card = dict(base)
        if mc_vto is not None:
            card["vto"] = Param(mc_vto, scale=float(base["type"]), offset=base["vto"])
        if mc_kp is not None:
            card["kp"] = Param(mc_kp, scale=base["kp"])
        if generated:
            c.VA("X_" + nm, "va_mos1l", (d, g, s, b), w=W, l=L, **{k: (float(v) if not isinstance(v, Param) else v) for k, v in card.items()})
        else:
            c.MOS1("X_" + nm, d, g, s, b, card, w=W, l=L)
    c.C("CQ", "Q_tmp", "0", 1.7205e-13)
    if not meyer:
        for (nm, d, g, s, b, W, L) in DFF_FETS:
            cg = lumped_gate_cap(W, L)
            c.C("Cgs_" + nm, g, s, cg)
            c.C("Cgd_" + nm, g, d, cg)
    return c


def dff_circuit_bsim4(vdd=None):
    """The same flip-flop with the reference's own BSIM4 text: every MOSFET an instance of ``sp_bsim4v8`` (models/VADistillerModels.jl/va/
    bsim4v8.va through the generator, csrc/va_generated_ext.hpp) on its default card -- SURVEY.md section 8d's secondary model for config 3
    ("bsim4v8 defaults level=14"; the gf180 PDK cards are not in the reference).  Only type and W / L are given; the model's own intrinsic
    and overlap charges load the nodes (no lumped gate capacitors).  ``vdd``: a number fixes the supply (no sweep parameter)."""
    c = Circuit("gf180 dffnq_4 test bench (sp_bsim4v8, default card)")
    v = Param("vdd") if vdd is None else float(vdd)
    amp = Param("vdd", scale=1.0 / 5.0) if vdd is None else float(vdd) / 5.0
    c.V("VVDD", "VDD", "0", dc=v)
    c.V("VVSS", "VSS", "0", dc=0.0)
    c.V("VQ", "Q", "Q_tmp", dc=0.0)
    c.V("VNW", "VNW", "VDD", dc=0.0)
    c.V("VPW", "VPW", "VSS", dc=0.0)
    c.V("VCLKN", "CLKN", "0", dc=(Param("vdd", scale=CLKN_PWL[1][0] / 5.0) if vdd is None else CLKN_PWL[1][0] * amp), wave=("pwl",) + CLKN_PWL, scale=amp)
    c.V("VD", "D", "0", dc=(Param("vdd", scale=D_PWL[1][0] / 5.0) if vdd is None else D_PWL[1][0] * amp), wave=("pwl",) + D_PWL, scale=amp)
    for (nm, d, g, s, b, W, L) in DFF_FETS:
        c.VA("X_" + nm, "sp_bsim4v8", (d, g, s, b), type=1.0 if nm.startswith("tn") else -1.0, w=W, l=L)
    c.C("CQ", "Q_tmp", "0", 1.7205e-13)
    return c


def inverter_circuit():
    """Single CMOS inverter transient of benchmarks/inverter_performance_bench.jl
    (benchmark_common.jl:82-105): nfet W=0.36u L=0.6u, pfet W=0.495u L=0.5u, VDD = 5 V, CQ = 1 fF,
    input PWL 0->5 V in 10 ns at 100/200/300 ns."""
    c = Circuit("gf180 inverter (synthetic level-1 cards)")
    ts = [0.0, 100e-9, 110e-9, 200e-9, 210e-9, 300e-9, 310e-9]
    ys = [0.0, 0.0, 5.0, 5.0, 0.0, 0.0, 5.0]
    c.V("VVDD", "VDD", "0", dc=Param("vdd"))
    c.V("VD", "D", "0", dc=0.0, wave=("pwl", ts, ys), scale=Param("vdd", scale=0.2))
    c.MOS1("X_tn", "Q", "D", "0", "0", NFET_06V0, w=0.36e-6, l=0.6e-6)
    c.MOS1("X_tp", "Q", "D", "VDD", "VDD", PFET_06V0, w=0.495e-6, l=0.5e-6)
    c.C("CQ", "Q", "0", 1e-15)
    for nm, d, g, s, W, L in (("tn", "Q", "D", "0", 0.36e-6, 0.6e-6), ("tp", "Q", "D", "VDD", 0.495e-6, 0.5e-6)):
        cg = lumped_gate_cap(W, L)
        c.C("Cgs_" + nm, g, s, cg)
        c.C("Cgd_" + nm, g, d, cg)
    return c


def corner_grid(n_vdd=32, n_temp=32, vdd=(4.5, 5.5), temp=(-40.0, 125.0)):
    """The 1024-point Vdd x temp corner grid of BASELINE.json config 4, ProductSweep order (Vdd fastest)."""
    from .api import ProductSweep
    return ProductSweep(vdd=np.linspace(vdd[0], vdd[1], n_vdd), temp=np.linspace(temp[0], temp[1], n_temp))


DFF_TSPAN = (0.0, 7e-7)
DFF_Q_PINS = ((150e-9, 0.0), (250e-9, 0.0), (450e-9, 5.0), (550e-9, 5.0), (700e-9, 5.0))  # test/gf180_dff.jl:29-33
