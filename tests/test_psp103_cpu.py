"""PSP103 (BASELINE.json config 5) on the CPU side: the committed fixtures tests/golden/psp103_*.npz are what
tools/make_psp103_fixtures.py produces from the reference's model text today, the structure carries the numbers the reference
documents, and the oracle passes the reference's own PSP103 tests.  Needs the model source (/root/reference or CADNIP_VA_PATH);
the fixture-only checks run everywhere."""
import os
import sys

import numpy as np
import pytest

from cadnip_jl_amd import structure as S, va

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HAVE_SOURCE = all(va.external_source(fn, sd) is not None for _, fn, sd in va.EXTERNAL)
needs_source = pytest.mark.skipif(not HAVE_SOURCE, reason="psp103.va is not at hand (no reference checkout, CADNIP_VA_PATH unset)")


def test_ring_fixture_carries_the_reference_layout():
    """doc/ring_oscillator_investigation.md:22-26: 371 unknowns, C with 1296 stored entries; 8 internal nodes per device with
    hierarchical names (test/mna/psp103_integration.jl:160-172); one branch current per executed V(a,b) <+ 0 of the model's
    CollapsableR macro (vasim.jl:2311-2395) and 5 charge unknowns per device make up the rest: 10 + 1 + 18 (8 + 7 + 5) = 371."""
    st, x = S.load_structure(os.path.join(GOLD, "psp103_ring.npz"))
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits) == (371, 154, 127, 90, 0)
    assert int((np.diff(st.c_ptr) > 0).sum()) == 1296
    internal = [nm for nm in st.node_names if "PSP103VA" in nm]
    assert len(internal) == 18 * 8 and len(set(internal)) == len(internal)
    for inst in ("xu1_xmn_nm", "xu1_xmp_nm", "xu9_xmn_nm", "xu9_xmp_nm"):
        assert sum(nm.startswith(inst + "_PSP103VA_") for nm in internal) == 8
    assert st.current_names[0] == "I_vdd" and st.current_names[1] == "xu1_xmp_nm_I_V_G_GP" and len(st.current_names) == 1 + 18 * 7
    assert x["U"].shape == (5, 371) and x["G"].shape == (5, st.nnz) and np.isfinite(x["G"]).all() and np.isfinite(x["b"]).all()
    blk = next(b for b in st.blocks if b.type == "VA:PSP103VA")
    assert blk.count == 18 and np.all(blk.ipar[0] == len(va.MODEL_FILES)) and np.all(blk.ipar[2] == 0x7F)


@needs_source
@pytest.mark.parametrize("name", ["nmos_defaults", "nmos_card", "ring", "bsim4_nmos", "bsim4_dff", "resistor", "capacitor", "diode", "diode_rs", "bjt",
                                  "jfet1", "mes1", "jfet2", "mos1", "mos2", "mos3", "mos6", "mos9", "bsim3v3", "bsim4v8", "bsimcmg_nmos", "juncap200", "inductor", "vdmos", "nlvcr", "noise_diode", "noise_bjt", "tm_1d", "tm_1d_interior", "tm_2d"])
def test_committed_fixtures_are_current(name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_psp103_fixtures as mk
    st, extra = mk.build(name)
    st0, x0 = S.load_structure(mk.fixture_path(name))
    assert st.signature() == st0.signature()
    for k in ("U", "G", "C", "b") + (("dc_x", "dc_ok") if "dc_ok" in extra else ()):
        assert np.array_equal(extra[k], x0[k], equal_nan=True), k
    for i in range(int(x0["n_packed"][0])):
        assert np.array_equal(extra["packed%d" % i], x0["packed%d" % i])


@needs_source
def test_oracle_passes_the_reference_psp103_dc_tests():
    """test/mna/psp103_integration.jl:40-122 on the oracle: V(d) = 1.2, V(g) = 0.6, |Id| in (100 uA, 1 mA) with the default card
    and in (10 uA, 10 mA) with the partial VACASK card."""
    for name, lo, hi in (("nmos_defaults", 100e-6, 1e-3), ("nmos_card", 10e-6, 10e-3)):
        st, x = S.load_structure(os.path.join(GOLD, "psp103_%s.npz" % name))
        u = x["dc_x"]
        assert abs(u[st.index_of("d")] - 1.2) < 1e-6 and abs(u[st.index_of("g")] - 0.6) < 1e-6
        assert lo < abs(u[st.index_of("I_Vds")]) < hi
        assert sum("PSP103VA" in nm for nm in st.node_names) == 8


@needs_source
def test_generated_external_sources_are_current():
    """csrc/va_generated_ext.hpp and csrc/va_ext/<module>.hip (committed: the library must build where the model sources are absent) are
    what the generator writes from the sources today."""
    from cadnip_jl_amd.va import hipgen, frontend
    mods = [frontend.parse_file(va.external_source(fn, sd)) for _, fn, sd in va.EXTERNAL]
    csrc = os.path.join(ROOT, "cadnip.jl_amd", "csrc")
    assert open(os.path.join(csrc, "va_generated_ext.hpp")).read() == hipgen.generate_ext_header(mods)
    assert sorted(os.listdir(os.path.join(csrc, "va_ext"))) == sorted(m.name + ".hip" for m in mods)
    for m in mods:
        assert open(os.path.join(csrc, "va_ext", m.name + ".hip")).read() == hipgen.generate_ext_unit(m), m.name


def test_tier6_fixtures_carry_the_reference_windows():
    """test/mna/vadistiller_integration.jl Tier 6: the oracle's DC solution of every circuit lies in the window the reference asserts."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_psp103_fixtures as mk
    for name, (_, probe, lo, hi) in mk.TIER6.items():
        st, x = S.load_structure(mk.fixture_path(name))
        assert lo < x["dc_x"][st.index_of(probe)] < hi, name
        assert bool(x["dc_ok"][0]) == (name not in ("mos3", "mos9")), name


def test_bsim4_dff_fixture_layout():
    """The flip-flop on the reference's bsim4v8.va (default card): 30 devices, each with 2 surviving internal nodes, 2 branch currents of
    executed V(a,b) <+ 0 statements, 4 charge states and 9 $limit unknowns; 31 derivative directions -> 32 lanes per device."""
    st, x = S.load_structure(os.path.join(GOLD, "bsim4_dff.npz"))
    assert (st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits) == (535, 18 + 60, 7 + 60, 120, 270)
    blk = next(b for b in st.blocks if b.type == "VA:sp_bsim4v8")
    assert blk.count == 30 and np.all(blk.ipar[0] == len(va.MODEL_FILES) + 1)
    if HAVE_SOURCE:
        from cadnip_jl_amd.va import hipgen
        m = va.get("sp_bsim4v8")[1]
        assert (m.n_nodes, m.n_sites, hipgen.tl_lanes(m)) == (13, 18, 32) and hipgen.tl_lanes(va.get("PSP103VA")[1]) == 16


@needs_source
def test_psp103_passes_the_gummel_symmetry_test():
    """A pin on the interpreted model that does not come from this repository's parser: PSP is a surface-potential model with a symmetric
    linearisation and passes the Gummel symmetry test by construction (the property the model's authors publish it for) -- with
    V(d) = +Vx, V(s) = -Vx at fixed gate and bulk the drain current is an ODD function of Vx, through the origin with a continuous slope.
    Nothing in the parser, the preprocessor (the model is five include files of macros), the branch analysis or the dual arithmetic
    knows that; a mis-read expression breaks it.  Checked on the oracle's DC solutions of the reference's NMOS test device
    (test/mna/psp103_integration.jl:40-62: W = 10 u, L = 1 u, default card): Id(+Vx) = -Id(-Vx) to 1e-9 relative for Vx from 1 mV to
    0.3 V, Id(0) = 0, and Id / Vx is the same to 0.5 % at 1 mV and 2 mV (no kink at the origin)."""
    from cadnip_jl_amd import netlist
    from oracle import mna_ref as M
    from oracle.netlist_ref import make_builder

    def drain_current(vx, vg=0.8):
        deck = ("* PSP103VA Gummel symmetry\n.model nch psp103va type=1\nM1 d g s 0 nch W=10u L=1u\nVd d 0 DC %r\nVs s 0 DC %r\nVg g 0 DC %r\n" % (vx, -vx, vg))
        circ = netlist.read_spice(deck)[0]
        sol = M.solve_dc(make_builder(circ.to_dicts({})), {}, M.MNASpec(mode="dcop", temp=27.0))
        assert sol.converged
        return -float(sol["I_Vd"]), -float(sol["I_Vs"])

    i0d, i0s = drain_current(0.0)
    assert abs(i0d) < 1e-15 and abs(i0s) < 1e-15
    slopes = []
    for vx in (1e-3, 2e-3, 2e-2, 0.1, 0.3):
        idp, isp = drain_current(vx)
        idm, ism = drain_current(-vx)
        assert idp > 0 and abs(idp + idm) <= 1e-9 * abs(idp), (vx, idp, idm)
        assert abs(idp + isp) <= 1e-9 * abs(idp) + 1e-15 and abs(isp - idm) <= 1e-9 * abs(idp)       # what enters the drain leaves the source; swapping the terminals swaps the currents
        slopes.append(idp / vx)
    assert abs(slopes[0] - slopes[1]) <= 5e-3 * slopes[0], slopes
