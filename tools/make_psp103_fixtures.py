"""Golden fixtures for BASELINE.json's config 5 (PSP103 ring oscillator) and the reference's PSP103 DC tests.

PSP103's Verilog-A source is third-party text inside the reference (models/PSPModels.jl/va) and is not copied into this
repository, so it is absent on the GPU box.  What travels instead -- written by this script, run in the build container where
/root/reference exists -- is DATA:

  tests/golden/psp103_<case>.npz     the discovered structure (cadnip_jl_amd.structure.save_structure), the packed
                                     per-instance parameter blocks, K states u with the ORACLE's G, C, b at each of them
                                     (oracle/va_ref.py interpreting the model text on the oracle's own duals through
                                     oracle/mna_ref.py's literal fast_rebuild!), and for the DC cases the oracle's DC solution.

Also `bsim4_nmos` and `bsim4_dff` (tests/golden/bsim4_*.npz): the reference's bsim4v8.va on its default card -- one NMOS, and the benchmark flip-flop
with 30 of them at 1.8 V (SURVEY.md section 8d's secondary model for config 3).

Cases: `nmos_defaults`, `nmos_card` -- test/mna/psp103_integration.jl:40-122 (the reference asserts |Id| in (100 uA, 1 mA) and
(10 uA, 10 mA)); `ring` -- benchmarks/vacask/ring/cedarsim/runme.sp + models.inc (kept as data fixtures psp103_ring.sp /
psp103_models.inc), n = 371 (doc/ring_oscillator_investigation.md:22).

    python tools/make_psp103_fixtures.py            (needs /root/reference or CADNIP_VA_PATH)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cadnip_jl_amd as cj                                   # noqa: E402
from cadnip_jl_amd import netlist, structure as S            # noqa: E402
from oracle import mna_ref as M                              # noqa: E402
from oracle.netlist_ref import make_builder                  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

NMOS_DEFAULTS = """* PSP103VA NMOS IV test with defaults
.model nch psp103va type=1
M1 d g 0 0 nch W=10u L=1u
Vds d 0 DC 1.2
Vgs g 0 DC 0.6
"""
# test/mna/psp103_integration.jl:67-106: a subset of the VACASK card
NMOS_CARD = """* PSP103VA with full model parameters
.model nch psp103va type=1 tr=27.0 vfbo=-1.1 vfbl=0 vfbw=0 stvfbo=5.0e-4 toxo=1.5e-9 epsroxo=3.9 nsubo=3.0e+23 nsubw=0 wseg=1.5e-10
+ npck=1.0e+24 npckw=0 wsegp=0.9e-8 lpck=5.5e-8 lpckw=0 fol1=2.0e-2 fol2=5.0e-6 facneffaco=0.8 gfacnudo=0.1 npo=1.5e+26 npl=10.0e-18
+ cto=5.0e-15 ctl=4.0e-2 ctlexp=0.6 toxovo=1.5e-9 toxovdo=2.0e-9 lov=10.0e-9 lovd=0 wot=0 thesato=0.5 mueo=0.5 mue=0.5
M1 d g 0 0 nch W=10u L=1u
Vds d 0 DC 1.2
Vgs g 0 DC 0.6
"""


def ring_deck():
    deck = open(os.path.join(GOLD, "psp103_ring.sp")).read()
    inc = open(os.path.join(GOLD, "psp103_models.inc")).read()
    return '.include "models.inc"\n' + deck.split("\n", 1)[1], {"models.inc": inc}     # (the first line is the title)


BSIM4_NMOS = """* sp_bsim4v8 (models/VADistillerModels.jl/va/bsim4v8.va) NMOS on its default card
.model nch sp_bsim4v8 type=1
M1 d g 0 0 nch W=1u L=0.5u
Vds d 0 DC 1.2
Vgs g 0 DC 0.8
"""


# test/mna/vadistiller_integration.jl:185-775 ("Tier 6: Full VADistiller Models (from file)"): one small DC circuit per model of
# models/VADistillerModels.jl, with the bounds the reference asserts on the solution.  name -> (deck, probe net, lower, upper)
def _stage(vdd, vg, rd, inst):
    return "Vdd vdd 0 DC %s\nVg gate 0 DC %s\nRd vdd drain %s\n%s\n" % (vdd, vg, rd, inst)


TIER6 = {
    "resistor": ("* sp_resistor divider (:191-213)\nV1 vcc 0 DC 5\nX1 vcc mid sp_resistor resistance=1000\nX2 mid 0 sp_resistor resistance=1000\n", "mid", 2.5 - 1e-9, 2.5 + 1e-9),
    "capacitor": ("* sp_capacitor DC (:215-238)\nV1 vcc 0 DC 5\nR1 vcc mid 1k\nX1 mid 0 sp_capacitor capacitance=1u\n", "mid", 5.0 - 1e-6, 5.0 + 1e-6),
    "inductor": ("* sp_inductor DC (:240-261)\nV1 vcc 0 DC 5\nR1 vcc mid 1k\nX1 mid 0 sp_inductor inductance=1m\n", "mid", -0.01, 0.01),
    "vdmos": ("* sp_vdmos (:693-712)\nVdd vdd 0 DC 10\nVg gate 0 DC 5\nRd vdd drain 100\nXm drain gate 0 0 0 sp_vdmos vto=2 kp=0.5\n", "drain", 0.0, 10.0),
    "diode": ("* sp_diode (:268-283)\nV1 vcc 0 DC 1\nR1 vcc diode_a 1k\nX1 diode_a 0 sp_diode\n", "diode_a", 0.6, 0.7),
    "diode_rs": ("* sp_diode with series resistance (:322-346)\nV1 vcc 0 DC 1\nR1 vcc diode_a 1k\nX1 diode_a 0 sp_diode rs=10\n", "diode_a", 0.6, 0.71),
    "bjt": ("* sp_bjt (:375-398)\nV1 vcc 0 DC 5\nV2 vb 0 DC 0.7\nRb vb base 10k\nRc vcc collector 1k\nXq collector base 0 0 sp_bjt bf=100 is=1e-15\n", "collector", 0.0, 5.0),
    "jfet1": ("* sp_jfet1 (:427-447)\n" + _stage(10, 0, "1k", "Xj drain gate 0 sp_jfet1 vt0=-2 beta=1m"), "drain", 0.0, 10.0),
    "mes1": ("* sp_mes1 (:449-469)\n" + _stage(5, 0, 500, "Xj drain gate 0 sp_mes1 vt0=-1 beta=2.5m"), "drain", 0.0, 5.0),
    "jfet2": ("* sp_jfet2 (:471-491)\n" + _stage(10, 0, "1k", "Xj drain gate 0 sp_jfet2 vto=-2 beta=1m"), "drain", 0.0, 10.0),
    "mos1": ("* sp_mos1 (:498-518)\n" + _stage(5, 2, "1k", "Xm drain gate 0 0 sp_mos1 l=1u w=10u vto=0.7 kp=1e-4"), "drain", 0.0, 5.0),
    "mos2": ("* sp_mos2 (:520-540)\n" + _stage(5, 2, "1k", "Xm drain gate 0 0 sp_mos2 l=1u w=10u vto=0.7 kp=1e-4"), "drain", 0.0, 5.0),
    "mos3": ("* sp_mos3 (:542-562)\n" + _stage(5, 2, "1k", "Xm drain gate 0 0 sp_mos3 l=1u w=10u vto=0.7 kp=1e-4"), "drain", 0.0, 5.0),
    "mos6": ("* sp_mos6 (:564-584)\n" + _stage(5, 2, "1k", "Xm drain gate 0 0 sp_mos6 l=1u w=10u vto=0.7 u0=600 tox=10n"), "drain", 0.0, 5.0),
    "mos9": ("* sp_mos9 (:586-606)\n" + _stage(5, 2, "1k", "Xm drain gate 0 0 sp_mos9 l=1u w=10u vto=0.7 kp=1e-4"), "drain", 0.0, 5.0),
    "bsim3v3": ("* sp_bsim3v3 (:716-742)\n" + _stage(1.8, 1.0, "1k", "Xm drain gate 0 0 sp_bsim3v3 l=100n w=1u"), "drain", 0.0, 1.8),
    "bsim4v8": ("* sp_bsim4v8 (:756-775)\n" + _stage(1.0, 0.5, "1k", "Xm drain gate 0 0 sp_bsim4v8 l=100n w=1u"), "drain", 0.9, 1.0),
}


# the other model packages of the reference: models/CMCModels.jl (BSIM-CMG 107, one fin on the default card) and the JUNCAP200 diode of
# models/PSPModels.jl.  No test of the reference that runs holds numbers for them (test/bsimcmg is legacy): the oracle's interpreter is the pin.
OTHER = {
    "bsimcmg_nmos": "* bsimcmg NMOS, default card\n.model nfin bsimcmg\nM1 d g 0 0 nfin\nVds d 0 DC 0.8\nVgs g 0 DC 0.6\n",
    "nlvcr": "* test/ddx.jl:38-73: NLVCR(R=2) between vcc (5 V), vg (3 V) and ground: I = 2 R V(d,s) V(g,s) = 60 A\nV1 vcc 0 DC 5\nV2 vg 0 DC 3\nX1 vcc vg 0 NLVCR R=2\n",
    # test/mna/table_model.jl:83-96 (fixtures tm_1d.scs / tm_2d.scs): V1 = 1 V across the table-model device, I_V1 = -g(wl[, T])
    "tm_1d": "* $table_model 1-D, wl = 1.55 (a sample): I_V1 = -0.02\nV1 p 0 DC 1\nX1 p 0 TMRoundTrip wl=1.55\n",
    "tm_1d_interior": "* $table_model 1-D, wl = 1.545 (between samples): I_V1 = -0.015\nV1 p 0 DC 1\nX1 p 0 TMRoundTrip wl=1.545\n",
    "tm_2d": "* $table_model 2-D, wl = 1.55, T = 25: I_V1 = -(2 wl + 3 T + 5)\nV1 p 0 DC 1\nX1 p 0 TM2D wl=1.55 T=25\n",
    "juncap200": "* JUNCAP200 diode behind 100 Ohm\nV1 a 0 DC 0.5\nR1 a k 100\nX1 k 0 juncap200\n",
}


# test/noise.jl:161-189: a forward-biased sp_diode loaded by a resistor (shot noise against the resistor's thermal noise) and an sp_bjt in the
# active region (three parasitic-resistance thermal sources, collector and base shot noise, base flicker noise).  name -> (deck, output, freqs)
NOISE = {
    "noise_diode": ("* diode_shot (test/noise.jl:14-19)\nV1 in 0 DC 5\nR1 in out 10k\nXd1 out 0 sp_diode is=1e-14 rs=0\n", "out", [1e2, 1e4]),
    "noise_bjt": ("* bjt_noise (test/noise.jl:21-27)\nVcc vcc 0 DC 5\nVb b 0 DC 0.7\nRc vcc c 4.7k\nXq1 c b 0 0 sp_bjt bf=100 is=1e-15 rb=100 re=1 rc=10 kf=1e-12 af=1\n", "c", [1e1, 1e2]),
}
NOISE_KINDS = ("thermal", "shot", "white", "flicker")


def fixture_path(name):
    return os.path.join(GOLD, "%s%s.npz" % ("" if name.startswith("bsim4_") else "vad_" if name in TIER6 else "va_" if name in OTHER or name in NOISE else "psp103_", name))


def cases():
    """name -> (deck text or Circuit, includes, mode)"""
    from cadnip_jl_amd import benchmarks as bm
    out = {"nmos_defaults": (NMOS_DEFAULTS, {}, "dcop"), "nmos_card": (NMOS_CARD, {}, "dcop"), "ring": ring_deck() + ("tran",),
           "bsim4_nmos": (BSIM4_NMOS, {}, "dcop"), "bsim4_dff": (bm.dff_circuit_bsim4(vdd=1.8), {}, "tran")}
    out.update({k: (v[0], {}, "dcop") for k, v in TIER6.items()})
    out.update({k: (v, {}, "dcop") for k, v in OTHER.items()})
    out.update({k: (v[0], {}, "dcop") for k, v in NOISE.items()})
    return out


def states(st, K, seed, vmax):
    """K probe states: zero, then random node voltages in [-0.1, 1.1] vmax, small branch currents, charges of either sign."""
    rng = np.random.default_rng(seed)
    U = np.zeros((K, st.n))
    a, b, c = st.n_nodes, st.n_nodes + st.n_currents, st.n_nodes + st.n_currents + st.n_charges
    for k in range(1, K):
        U[k, :a] = (rng.random(a) * 1.2 - 0.1) * vmax
        U[k, a:b] = (rng.random(b - a) - 0.5) * 1e-3
        U[k, b:c] = (rng.random(c - b) - 0.5) * 1e-2
    return U


def build(name, K=5):
    deck, includes, mode = cases()[name]
    circ = deck if not isinstance(deck, str) else netlist.read_spice(deck, includes=includes)[0]
    st = cj.discover(circ, {})
    packed = cj.pack_params(st, circ, {}, np.array([27.0]), 1, gmin=1e-12)
    bld = make_builder(circ.to_dicts({}))
    spec = M.MNASpec(mode=mode, temp=27.0)
    ctx = M.build_with_detection(bld, {}, spec)
    cs = M.compile_structure(bld, {}, spec, ctx=ctx)
    ws = M.create_workspace(cs, ctx=ctx)
    assert (st.n, st.node_names, st.current_names, st.charge_names) == (cs.n, ctx.node_names, ctx.current_names, ctx.charge_names)
    assert np.array_equal(st.ref_colptr, cs.colptr) and np.array_equal(st.ref_rowval, cs.rowval)
    U = states(st, K, 20261004, 1.8 if name == "bsim4_dff" else 1.2)
    if name in TIER6 or name in OTHER or name in NOISE:      # (volts of the circuit's own scale; limit unknowns near their probes)
        U = states(st, K, 20261004, 2.0)
        U[1:, st.n - st.n_limits:] = np.random.default_rng(7).random((K - 1, st.n_limits)) * 1.4 - 0.2
    T = np.array([0.0, 0.0, 1.5e-9, 3e-9, 7e-9][:K])
    Gs, Cs, bs = [], [], []
    for u, t in zip(U, T):
        M.fast_rebuild(ws, u, float(t))
        Gs.append(cs.G.data.copy()); Cs.append(cs.C.data.copy()); bs.append(ws.dctx.b.copy())
    extra = dict(U=U, T=T, G=np.array(Gs), C=np.array(Cs), b=np.array(bs), mode=np.frombuffer(mode.encode(), dtype=np.uint8),
                 n_packed=np.array([len(packed)]))
    for i, p in enumerate(packed):
        extra["packed%d" % i] = p
    if mode == "dcop":
        sol = M.solve_dc(bld, {}, spec)
        # (the reference's solve_dc returns whatever its fallback chain ends with; its Tier 6 tests assert bounds, not convergence: sp_mos3 (and sp_mos9, the same equations)
        # on its default card takes sqrt(0 * dual) in saturation -- a NaN partial under ForwardDiff -- and ends unconverged there too)
        assert sol.converged or name in ("mos3", "mos9"), name
        extra["dc_x"] = np.asarray(sol.x, dtype=float)
        extra["dc_ok"] = np.array([1 if sol.converged else 0])
    if name in NOISE:
        # the oracle's noise! at its own DC point: the registered sources (system indices, 0 = ground) and the output PSD
        _, output, freqs = NOISE[name]
        onoise, contributions, _, _ = M.noise(bld, {}, spec, output, freqs)
        nctx = M.MNAContext()
        bld({}, spec, 0.0, x=M.ZERO_VECTOR, ctx=nctx)
        nctx.reset_for_restamping()
        nctx.noise = []
        bld({}, spec, 0.0, x=sol.x, ctx=nctx)
        extra["noise_freqs"], extra["noise_onoise"] = np.array(freqs, dtype=float), onoise
        extra["noise_p"] = np.array([nctx.resolve_index(q[0]) for q in nctx.noise])
        extra["noise_n"] = np.array([nctx.resolve_index(q[1]) for q in nctx.noise])
        extra["noise_kind"] = np.array([NOISE_KINDS.index(q[2]) for q in nctx.noise])
        extra["noise_a"], extra["noise_b"] = np.array([q[3] for q in nctx.noise]), np.array([q[4] for q in nctx.noise])
        extra["noise_names"] = np.frombuffer(",".join(q[5] for q in nctx.noise).encode(), dtype=np.uint8)
        extra["noise_output"] = np.frombuffer(output.encode(), dtype=np.uint8)
    return st, extra


def main():
    for name in (sys.argv[1:] or cases()):
        st, extra = build(name)
        path = fixture_path(name)
        S.save_structure(st, path, **extra)
        print("%-14s n = %d (nodes %d, currents %d, charges %d)  nnz %d  ->  %s (%.1f KB)" % (
            name, st.n, st.n_nodes, st.n_currents, st.n_charges, st.nnz, os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
