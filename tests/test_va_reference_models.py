"""The Verilog-A front end reads the reference's own model files (counterpart of the reference's VA
lowering, src/vasim.jl:2993-3985).  The files are third-party sources inside the reference and are NOT copied: these tests
read them where they lie and skip when /root/reference is absent (the GPU box).

What is pinned here:
  * every VADistiller model and PSP103 / JUNCAP200 parses and passes the static analysis, with the structural numbers the
    reference documents (mos1: 4 + 2 nodes, 4 $limit probe branches x 2 sites; bsim4v8: 4 + 9 nodes; PSP103: 4 + 8 nodes,
    test/mna/psp103_integration.jl:160) -- or is refused for a stated reason (branch-current unknowns);
  * the reference's mos1.va, interpreted statement by statement on the oracle's dual numbers (oracle/va_ref.py), stamps
    EXACTLY what the hand transcription of the same file stamps (oracle/va_mos1_ref.py, the oracle behind the sp_mos1
    kernels): G, C, b and limit_w bit for bit, over cards that reach every section of the model (Meyer charges, junction
    charges, series resistances = genuine internal nodes, PMOS, multiplicity, temperature, initjct);
  * resistor.va / capacitor.va stamp the closed forms.
"""
import os

import numpy as np
import pytest

from cadnip_jl_amd.va import frontend as F, host_eval
from cadnip_jl_amd import benchmarks as bm
from oracle import mna_ref as M, va_ref
from oracle.mna_ref import MNAContext, ZERO_VECTOR
from oracle.va_mos1_ref import Mos1Model, stamp_mos1

VA_DIR = "/root/reference/models/VADistillerModels.jl/va"
PSP_DIR = "/root/reference/models/PSPModels.jl/va"
pytestmark = pytest.mark.skipif(not os.path.isdir(VA_DIR), reason="the reference checkout is not present")

# file -> (nodes, branches, reactive branches, $limit sites, node-collapse statements)
EXPECTED = {
    "resistor.va": (2, 2, 0, 0, 0), "capacitor.va": (2, 2, 2, 0, 0), "diode.va": (3, 3, 2, 2, 1),
    "jfet1.va": (5, 5, 3, 4, 2), "jfet2.va": (8, 8, 6, 4, 5), "mes1.va": (5, 5, 3, 4, 2),
    "mos1.va": (6, 6, 4, 8, 2), "mos2.va": (6, 6, 4, 8, 2), "mos3.va": (6, 6, 4, 8, 2), "mos6.va": (6, 6, 4, 8, 2), "mos9.va": (6, 6, 4, 8, 2),
    "bsim3v3.va": (7, 7, 5, 8, 3), "bsim4v8.va": (13, 23, 9, 18, 9), "bjt.va": (11, 11, 9, 6, 9),
    "inductor.va": (2, 0, 0, 0, 1),        # V(br) <+ req*I(br)+veq on a named branch: no current branch at all, one branch-current unknown
    "vdmos.va": (10, 12, 5, 8, 7),         # five terminals (two thermal), V(tbr) <+ ... on the named thermal branch
}
# refused, with the reason
REFUSED = {os.path.join(PSP_DIR, "psp103_nqs.va"): "idt"}


def test_table_model_interpolator_numerics():
    """test/mna/table_model.jl:43-69: the 1-D and 2-D interpolators on the reference's fixture tables -- sample, interior, linear extrapolation
    below and above the range -- in the product's host evaluator and in the oracle's restatement."""
    from cadnip_jl_amd.va import table_model as T
    d = "/root/reference/test/mna/fixtures/table_model/"
    for f in (T.lookup, va_ref.table_model_value):
        assert f(d + "tm_1d.tbl", "1L;1", [1.55]) == pytest.approx(0.02) and f(d + "tm_1d.tbl", "1L;1", [1.545]) == pytest.approx(0.015)
        assert abs(f(d + "tm_1d.tbl", "1L;1", [1.53])) < 1e-12 and f(d + "tm_1d.tbl", "1L;1", [1.57]) == pytest.approx(0.04)
        for a, b in ((1.55, 25.0), (1.545, 22.5), (1.53, 15.0), (1.57, 35.0)):
            assert f(d + "tm_2d.tbl", "1L,1L;1", [a, b]) == pytest.approx(2 * a + 3 * b + 5)
        assert f(d + "tm_1d.tbl", "1C;1", [1.57]) == pytest.approx(0.03)
        with pytest.raises(ValueError):
            f(d + "tm_1d.tbl", "1E;1", [1.57])
    m = F.parse_file(d + "tm_1d.va")
    assert len(m.table_calls) == 1 and m.shape()[4] == len(m.params) + 3 + 1
    with pytest.raises(F.VAError):          # a look-up of a bias-dependent quantity has no host-side value
        F.parse_module('module x(a); electrical a; analog I(a) <+ $table_model(V(a), "t.tbl", "1L;1"); endmodule')


@pytest.mark.parametrize("fn", sorted(EXPECTED))
def test_reference_model_parses(fn):
    m = F.parse_file(os.path.join(VA_DIR, fn))
    assert (m.n_nodes, len(m.branches), sum(m.reactive), m.n_sites, len(m.shorts)) == EXPECTED[fn]
    assert len(m.ports) in (2, 3, 4, 5) and m.name.startswith("sp_")


@pytest.mark.parametrize("fn", sorted(REFUSED))
def test_unsupported_models_are_refused_with_a_reason(fn):
    with pytest.raises(F.VAError) as ei:
        F.parse_file(fn)
    assert REFUSED[fn] in str(ei.value)


def test_potential_contributions_of_inductor_and_vdmos():
    """inductor.va: `V(br) <+ req*I(br)+veq` with veq = ddt(L * I(br)) -- a named branch at the top level of the analog block: its current is
    allocated up front, I(br) reads it as a plain number, the value carries ddt() (vasim.jl:3253-3266, 3632-3640, 3669-3746).  vdmos.va: the
    thermal branch `V(tbr) <+ ...` likewise; its conditional V(t) <+ 0 / V(tc) <+ 0 tie terminals to ground through branch currents."""
    m = F.parse_file(os.path.join(VA_DIR, "inductor.va"))
    assert m.short_kind == ["named"] and m.short_reactive == [True] and m.vshorts == [0]
    assert m.short_current_name(0, "l1") == "l1_sp_inductor_I_br"
    assert m.shape()[:4] == (2 + 0 + 0 + 1, (4 + 2) * 1, 1, 1)      # unknowns, G slots, the one C slot (I,I), the one b slot
    assert m.program((), [], [True]) == [("G", 0, 0, 2), ("G", 1, 1, 2), ("G", 2, 2, 0), ("G", 3, 2, 1), ("b", 0, 2, None), ("C", 0, 2, 2)]
    v = F.parse_file(os.path.join(VA_DIR, "vdmos.va"))
    assert [v.short_kind[i] for i in v.vshorts] == ["named", "cond", "cond"] and v.short_current_name(0, "m1") == "m1_sp_vdmos_I_tbr"
    assert v.short_current_name(1, "m1") == "m1_I_V_t_0"


def test_other_model_packages_parse():
    """models/CMCModels.jl (BSIM-CMG 107: 905 parameters, 2 internal nodes with the series-resistance option compiled in) and the self-heating
    PSP103 (a fifth, thermal terminal; Temp() / Pwr() access functions)."""
    c = F.parse_file("/root/reference/models/CMCModels.jl/va/bsimcmg.va")
    assert (c.name, c.ports, c.n_internal, len(c.params), len(c.branches), sum(c.reactive)) == ("bsimcmg", ["d", "g", "s", "e"], 2, 905, 15, 9)
    t = F.parse_file(os.path.join(PSP_DIR, "psp103t.va"))
    assert (t.name, t.ports, t.n_internal, len(t.branches)) == ("PSP103TVA", ["D", "G", "S", "B", "DT"], 8, 19)


def test_psp103_and_juncap_parse_through_their_includes():
    m = F.parse_file(os.path.join(PSP_DIR, "psp103.va"))
    assert m.name == "PSP103VA" and m.ports == ["D", "G", "S", "B"]
    assert m.n_internal == 8                                   # test/mna/psp103_integration.jl:160: 8 internal nodes per device
    assert len(m.params) > 700 and len(m.branches) == 18 and sum(m.reactive) == 8
    j = F.parse_file(os.path.join(PSP_DIR, "juncap200.va"))
    assert j.ports == ["A", "K"] and len(j.branches) == 1 and j.reactive == [True]


def _systems(mod, card, mfactor, spec):
    par = host_eval.defaults(mod, card)

    def hand(params, spec, t, x=ZERO_VECTOR, ctx=None):
        ctx = ctx or MNAContext()
        nd = [ctx.get_node(n) for n in ("d", "g", "s", "b")]
        stamp_mos1(ctx, Mos1Model(**card), nd[0], nd[1], nd[2], nd[3], x, spec, "m1", mfactor=mfactor)
        return ctx

    def from_text(params, spec, t, x=ZERO_VECTOR, ctx=None):
        ctx = ctx or MNAContext()
        nd = [ctx.get_node(n) for n in ("d", "g", "s", "b")]
        va_ref.stamp_va(ctx, mod, nd, x, par, spec, "m1", mfactor=mfactor, given=set(card))
        return ctx
    out = []
    for b in (hand, from_text):
        ctx = M.build_with_detection(b, {}, spec)
        cs = M.compile_structure(b, {}, spec, ctx=ctx)
        out.append((cs, M.create_workspace(cs, ctx=ctx)))
    return out


def _mos1_rd_card():
    card = dict(type=1, tox=2e-8, nsub=1e16, u0=500.0, rd=20.0, rs=15.0, cj=2e-4, cjsw=1e-10, mj=0.4, mjsw=0.3, js=1e-6,
                cgso=2e-10, cgdo=2e-10, cgbo=1e-10, ld=5e-8, l=1e-6, w=10e-6, ad=1e-11, pd=2e-5, ps=2e-5)
    card["lambda"] = 0.05
    card["as"] = 1e-11
    return card


def _bench(card, **inst):
    c = dict(card)
    c.update(inst)
    return c


CARDS = {
    "level1_minimal": (dict(type=1, vto=0.7, kp=100e-6, w=20e-6, l=1e-6, **{"lambda": 0.01}), 1.0, 27.0),
    "pmos_minimal": (dict(type=-1, vto=-0.7, kp=50e-6, w=2e-6, l=1e-6), 1.0, 27.0),
    "series_r_meyer_junctions": (_mos1_rd_card(), 2.0, 27.0),
    "series_r_hot": (_mos1_rd_card(), 1.0, 110.0),
    "benchmark_nfet": (_bench(bm.NFET_06V0, w=0.36e-6, l=0.6e-6), 1.0, -40.0),
    "benchmark_pfet": (_bench(bm.PFET_06V0, w=0.495e-6, l=0.5e-6), 1.0, 125.0),
}


@pytest.mark.parametrize("name", list(CARDS))
@pytest.mark.parametrize("mode", ["tran", "dcop"])
def test_reference_mos1_text_equals_the_hand_transcription(name, mode):
    card, mfactor, temp = CARDS[name]
    mod = F.parse_file(os.path.join(VA_DIR, "mos1.va"))
    (csA, wsA), (csB, wsB) = _systems(mod, card, mfactor, M.MNASpec(mode=mode, temp=temp))
    assert csA.n == csB.n and csA.G.nnz == csB.G.nnz and csA.n_limits == csB.n_limits == 4
    rng = np.random.default_rng(7)
    for trial in range(5):
        u = (rng.random(csA.n) * 6 - 2.0) * (1.0 if trial else 0.0)
        for initjct in ((True, False) if trial == 0 else (False,)):
            wsA.dctx.initjct = wsB.dctx.initjct = initjct
            M.fast_rebuild(wsA, u, 1e-9)
            M.fast_rebuild(wsB, u, 1e-9)
            assert np.array_equal(csA.G.toarray(), csB.G.toarray()) and np.array_equal(csA.C.toarray(), csB.C.toarray())
            assert np.array_equal(wsA.dctx.b, wsB.dctx.b) and np.array_equal(wsA.dctx.limit_w, wsB.dctx.limit_w)


def test_reference_resistor_and_capacitor_text_stamp_the_closed_forms():
    spec = M.MNASpec(mode="tran", temp=27.0)
    for fn, card, which, want in (("resistor.va", dict(resistance=2e3), "G", 1.0 / 2e3), ("capacitor.va", dict(capacitance=3e-12), "C", 3e-12),
                                  ("resistor.va", dict(model_r=500.0, tc=1e-3, dtemp=10.0), "G", 1.0 / (500.0 * (1.0 + 1e-3 * 10.0)))):
        mod = F.parse_file(os.path.join(VA_DIR, fn))
        par = host_eval.defaults(mod, card)

        def b(params, spec, t, x=ZERO_VECTOR, ctx=None, mod=mod, par=par, card=card):
            ctx = ctx or MNAContext()
            va_ref.stamp_va(ctx, mod, [ctx.get_node("p"), ctx.get_node("n")], x, par, spec, "x1", given=set(card))
            return ctx
        ctx = M.build_with_detection(b, {}, spec)
        cs = M.compile_structure(b, {}, spec, ctx=ctx)
        ws = M.create_workspace(cs, ctx=ctx)
        M.fast_rebuild(ws, np.array([1.3, 0.2] + [0.0] * (cs.n - 2)), 0.0)
        G, C = cs.G.toarray(), cs.C.toarray()
        A = (G if which == "G" else C)[:2, :2]
        if which == "C" and cs.n > 2:
            # the two single-node branches I(pos) <+ ddt(q), I(neg) <+ ddt(-q) see q = C (V(pos) - V(neg)) against ONE node
            # voltage each, so the detection passes find Q / V_branch varying and take the charge-state form
            # (contrib.jl:214-257): eliminate the charge unknowns, C_eff = -C_nq G_qq^-1 G_qn
            A = -C[:2, 2:] @ np.linalg.solve(G[2:, 2:], G[2:, :2])
        assert np.allclose(A, want * np.array([[1.0, -1.0], [-1.0, 1.0]]), rtol=1e-12, atol=0.0), (fn, A)


def test_reference_bjt_text_is_a_gummel_poon_transistor():
    """bjt.va (11 nodes; V(sub) <+ 0 ties the substrate terminal to ground through a branch current, two internal-internal shorts): the
    oracle's interpretation in a common-emitter stage -- Vbe = 0.7 V, bf = 100, is at its default 1e-16 A -- gives the textbook numbers
    Ic = Is exp(Vbe / Vt), Ib = Ic / bf."""
    import math
    from oracle import devices_ref as D
    mod = F.parse_file(os.path.join(VA_DIR, "bjt.va"))
    assert mod.ports == ["c", "b", "e", "sub"] and len(mod.vshorts) == 4
    card = dict(bf=100.0)
    par = host_eval.defaults(mod, card)

    def b(params, spec, t, x=ZERO_VECTOR, ctx=None):
        ctx = ctx or MNAContext()
        c, bb, vcc = ctx.get_node("c"), ctx.get_node("b"), ctx.get_node("vcc")
        D.stamp_vsource(ctx, vcc, 0, 5.0, name="vcc")
        D.stamp_vsource(ctx, bb, 0, 0.7, name="vb")
        D.stamp_resistor(ctx, vcc, c, 1e3)
        va_ref.stamp_va(ctx, mod, [c, bb, 0, 0], x, par, spec, "q1", given=set(card))
        return ctx
    sol = M.solve_dc(b, {}, M.MNASpec(mode="dcop", temp=27.0))
    assert sol.converged
    x = dict(zip(sol.sys.node_names + sol.sys.current_names, sol.x))
    ic, ib = -x["I_vcc"], -x["I_vb"]
    vt = 1.380649e-23 * 300.15 / 1.602176634e-19
    assert ic == pytest.approx(1e-16 * math.exp(0.7 / vt), rel=1e-3) and ib == pytest.approx(ic / 100.0, rel=1e-3)
    assert x["c"] == pytest.approx(5.0 - 1e3 * ic, rel=1e-9) and x["q1_I_V_c_int_cx_int"] == pytest.approx(-ic, rel=1e-9)


def _device_compile(header_text, tmp_path, opt="-O1"):
    """hipcc --cuda-device-only on a translation unit that instantiates every generated stamp function through stamp_va, against
    the product's devices.hpp / va_runtime.hpp (symlinked next to the generated header, which they include by name)."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc is not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "pkg" / "csrc"
    src.mkdir(parents=True)
    os.symlink(os.path.join(root, "include"), tmp_path / "include")
    for h in ("devices.hpp", "va_runtime.hpp"):
        os.symlink(os.path.join(root, "cadnip.jl_amd", "csrc", h), src / h)
    (src / "va_generated.hpp").write_text(header_text)
    from cadnip_jl_amd.va import hipgen
    (src / "va_generated_ext.hpp").write_text(hipgen.generate_ext_header([]))     # no external model beside the ones under test
    (src / "tu.hip").write_text('#include <hip/hip_runtime.h>\n#include "devices.hpp"\nusing namespace cadnip;\n'
                                "__global__ void k_tu(DevCtx d, const double* u, double* S, double* lw) {\n"
                                "  SlotOut s{S, S + 100000, S + 200000, d.count, d.dev};\n  stamp_va(d, u, s, lw);\n}\n")
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", opt, "-std=c++17", "--cuda-device-only", "-c", "-o", str(tmp_path / "tu.o"), "tu.hip"],
                       cwd=src, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert os.path.getsize(tmp_path / "tu.o") > 1000


def test_reference_models_generate_hip_that_compiles_for_gfx950(tmp_path):
    """resistor, capacitor, diode and mos1 of the reference -> one generated header -> device code for gfx950.  (The numbers the
    generated functions produce are pinned module by module on the GPU for the modules that ship with the library --
    tests/test_gpu_parity.py, tests/test_gpu_va.py; va_feat.va covers the constructs only the reference's files use -- and
    the reference's mos1.va itself is pinned above through the interpreter both generators' outputs are compared with.)"""
    from cadnip_jl_amd.va import hipgen
    mods = [F.parse_file(os.path.join(VA_DIR, f)) for f in ("resistor.va", "capacitor.va", "diode.va", "mos1.va", "bjt.va")]
    text = hipgen.generate_header(mods)
    assert "stamp_va_sp_mos1" in text and "vaf_sp_mos1_DEVqmeyer" in text and "g_tox" in text
    _device_compile(text, tmp_path)


@pytest.mark.parametrize("which", ["bsim4v8", "psp103"])
def test_big_reference_models_generate_and_compile(which, tmp_path):
    """BSIM4 v8 (9 950 lines, 928 parameters, string version tests) and PSP103.4 (through its five include files, 782
    parameters) -- the models of BASELINE.json's config 5 and of SURVEY.md 8d's secondary DFF card: generated and compiled."""
    from cadnip_jl_amd.va import hipgen
    m = F.parse_file(os.path.join(VA_DIR, "bsim4v8.va") if which == "bsim4v8" else os.path.join(PSP_DIR, "psp103.va"))
    text = hipgen.generate_header([m])
    assert len(text.splitlines()) > 5000
    _device_compile(text, tmp_path, opt="-O1")      # (-O0 keeps every dual on the stack: the frame outgrows the 128 KB limit)
