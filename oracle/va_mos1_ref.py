"""TEST INFRASTRUCTURE (oracle) -- literal restatement of the ``sp_mos1`` Verilog-A
device as Cadnip stamps it.

Two layers, both literal:

* ``va_stamp`` follows the stamp! body that
  /root/reference/src/vasim.jl:2993-3985 (generate_mna_stamp_method_nterm) emits:
  internal-node allocation with short-circuit aliasing (:3523-3564), the ``$limit``
  preamble (:3097-3146), JacobianTag duals of width n_nodes + n_limit_sites
  (:3610-3626), per-branch G / C-or-charge-state / b stamps (:3319-3521) including
  the lim_rhs re-anchoring (:2957-2966) and the per-device detection-counter reset
  (:3926).
* ``Mos1Model.evaluate`` is the analog block of
  /root/reference/models/VADistillerModels.jl/va/mos1.va:640-1169 (setup, temp,
  load, residuals; noise is dead on this path, value_only.jl:177) evaluated with
  forward-mode duals exactly as the generated Julia does.

Branch stamp order: the reference iterates a Julia ``Dict`` of branches
(vasim.jl:3217,3284), whose order is hash-dependent and cannot be reproduced
without Julia; declaration order is used here.  Only the floating-point
summation order inside an nz entry depends on it.
"""
import math
import numpy as np

from .dual import Dual, CDual, va_ddt, val, dsqrt, dexp, dln, dabs, dmax, dmin, partials
from .mna_ref import CHARGE_SCALE, x_at


# -----------------------------------------------------------------------------------------
# limiter functions (mos1.va:474-636) -- evaluated on dual inputs, like the generated code
# -----------------------------------------------------------------------------------------
def DEVpnjlim(vnew, vold, vt, vcrit, limiting_applied):  # mos1.va:503-540
    limited = vnew
    if (vnew > vcrit) and (dabs(vnew - vold) > (vt + vt)):
        if vold > 0:
            arg = (vnew - vold) / vt
            if arg > 0:
                limited = vold + vt * dln(1 + arg)
                limiting_applied = 1
            else:
                limited = vold - vt * dln(1 - arg)
                limiting_applied = 1
        else:
            limited = vt * dln(vnew / vt)
            limiting_applied = 1
    elif vnew < 0:
        if vold > 0:
            arg = -1 * vold - 1
        else:
            arg = 2 * vold - 1
        if vnew < arg:
            limited = arg
            limiting_applied = 1
    return limited, limiting_applied


def DEVfetlim(vnew, vold, vto, limiting_applied):  # mos1.va:542-605
    vlimited = vnew
    vtsthi = dabs(2 * (vold - vto)) + 2
    vtstlo = dabs(vold - vto) + 1
    vtox = vto + 3.5
    delv = vnew - vold
    if vold >= vto:
        if vold >= vtox:
            if delv <= 0:
                if vlimited >= vtox:
                    if -delv > vtstlo:
                        vlimited = vold - vtstlo
                else:
                    vlimited = dmax(vnew, vto + 2)
            else:
                if delv >= vtsthi:
                    vlimited = vold + vtsthi
        else:
            if delv <= 0:
                vlimited = dmax(vnew, vto - 0.5)
            else:
                vlimited = dmin(vnew, vto + 4)
    else:
        if delv <= 0:
            if -delv > vtsthi:
                vlimited = vold - vtsthi
        else:
            vtemp = vto + 0.5
            if vnew <= vtemp:
                if delv > vtstlo:
                    vlimited = vold + vtstlo
            else:
                vlimited = vtemp
    if val(vlimited) != val(vnew):
        limiting_applied = 1
    return vlimited, limiting_applied


def DEVlimvds(vnew, vold, limiting_applied):  # mos1.va:607-635
    if vold >= 3.5:
        if vnew > vold:
            vlimited = dmin(vnew, (3 * vold) + 2)
        else:
            if vnew < 3.5:
                vlimited = dmax(vnew, 2)
            else:
                vlimited = vnew
    else:
        if vnew > vold:
            vlimited = dmin(vnew, 4)
        else:
            vlimited = dmax(vnew, -0.5)
    if val(vlimited) != val(vnew):
        limiting_applied = 1
    return vlimited, limiting_applied


def DEVqmeyer(vgs, vgd, vgb, von, _vdsat, phi, cox):  # mos1.va:401-465; returns (capgs, capgd, capgb)
    vdsat = _vdsat
    vgst = vgs - von
    vdsat = vdsat if vdsat > 0.025 else 0.025
    if vgst <= -phi:
        capgb = cox / 2
        capgs = 0
        capgd = 0
    elif vgst <= -phi / 2:
        capgb = -vgst * cox / (2 * phi)
        capgs = 0
        capgd = 0
    elif vgst <= 0:
        capgb = -vgst * cox / (2 * phi)
        capgs = vgst * cox / (1.5 * phi) + cox / 3
        vds = vgs - vgd
        if vds >= vdsat:
            capgd = 0
        else:
            vddif = 2.0 * vdsat - vds
            vddif1 = vdsat - vds
            vddif2 = vddif * vddif
            capgd = capgs * (1.0 - vdsat * vdsat / vddif2)
            capgs = capgs * (1.0 - vddif1 * vddif1 / vddif2)
    else:
        vds = vgs - vgd
        vdsat = vdsat if vdsat > 0.025 else 0.025
        if vdsat <= vds:
            capgs = cox / 3
            capgd = 0
            capgb = 0
        else:
            vddif = 2.0 * vdsat - vds
            vddif1 = vdsat - vds
            vddif2 = vddif * vddif
            capgd = cox * (1.0 - vdsat * vdsat / vddif2) / 3
            capgs = cox * (1.0 - vddif1 * vddif1 / vddif2) / 3
            capgb = 0
    return capgs, capgd, capgb


# -----------------------------------------------------------------------------------------
# model
# -----------------------------------------------------------------------------------------
MOS1_DEFAULTS = dict(
    l=0.0, w=0.0, ad=0.0, **{"as": 0.0}, pd=0.0, ps=0.0, nrd=1.0, nrs=1.0, temp=0.0, dtemp=0.0,
    type=1, vto=0.0, kp=2e-5, gamma=0.0, phi=0.6, rd=0.0, rs=0.0, cbd=0.0, cbs=0.0,
    pb=0.8, cgso=0.0, cgdo=0.0, cgbo=0.0, rsh=0.0, cj=0.0, mj=0.5, cjsw=0.0, mjsw=0.5, js=0.0, tox=0.0,
    ld=0.0, u0=600.0, fc=0.5, nsub=0.0, tpg=1, nss=0.0, tnom=0.0, **{"lambda": 0.0, "is": 1e-14})

PORTS = ("d", "g", "s", "b")
INTERNAL = ("d_int", "s_int")
ALL_NODES = PORTS + INTERNAL
# unique $limit probe branches in first-appearance order (mos1.va:919-922)
LIMIT_BRANCHES = (("g", "s_int"), ("d_int", "s_int"), ("b", "s_int"), ("b", "d_int"))
# $limit call sites: 4 x OldGet (mos1.va:919-922) then 4 x NewSet (mos1.va:976-979)
LIMIT_SITES = LIMIT_BRANCHES + LIMIT_BRANCHES
N_NODES = 6
WIDTH = N_NODES + len(LIMIT_SITES)
BRANCHES = ("d", "g", "s", "b", "d_int", "s_int")  # I(x) <+ ... to ground, mos1.va:1164-1169


class Mos1Model:
    def __init__(self, **given):
        for k in given:
            if k not in MOS1_DEFAULTS:
                raise KeyError("unknown sp_mos1 parameter %r" % k)
        self.given = dict(given)
        self.p = dict(MOS1_DEFAULTS)
        self.p.update(given)

    def param_given(self, name):
        return name in self.given

    # short-circuit conditions (mos1.va:716-721; vasim.jl:2723-2818)
    def sc_d(self):
        p = self.p
        return not (p["rd"] != 0 or (p["rsh"] != 0 and p["nrd"] != 0))

    def sc_s(self):
        p = self.p
        return not (p["rs"] != 0 or (p["rsh"] != 0 and p["nrs"] != 0))

    def evaluate(self, V, limit_site, spec, initjct, mfactor=1.0):
        """V: dict node name -> Dual.  limit_site(j, vnew, fn) implements the $limit lowering
        (vasim.jl:1258-1330).  Returns dict branch name -> I_branch (float | Dual | CDual)."""
        P = self.p
        given = self.param_given
        typ = P["type"]
        lam = P["lambda"]
        # global constants (mos1.va:381-393)
        CONSTroot2 = math.sqrt(2.0)
        CONSTKoverQ = 1.38064852e-23 / 1.6021766208e-19
        defad, defas, defl, defw, oldlimit = 0.0, 0.0, 1e-4, 1e-4, 0.0
        VACONST_tnom = spec.tnom + 273.15
        cpscale = 1.0
        T_K = spec.temp + 273.15  # $temperature (vasim.jl:1181)

        def Vb(a, b):
            return V[a] - V[b]

        # instance / model field assignments (mos1.va:640-690)
        MOS1l = P["l"] * cpscale if given("l") else 0.0
        MOS1w = P["w"] * cpscale if given("w") else 0.0
        MOS1drainArea = P["ad"] * cpscale * cpscale if given("ad") else 0.0
        MOS1sourceArea = P["as"] * cpscale * cpscale if given("as") else 0.0
        MOS1drainPerimiter = P["pd"] * cpscale if given("pd") else 0.0
        MOS1sourcePerimiter = P["ps"] * cpscale if given("ps") else 0.0
        MOS1tempGiven = given("temp")
        MOS1temp = P["temp"] + 273.15 if MOS1tempGiven else 0.0
        MOS1type = typ
        MOS1vt0 = P["vto"] if given("vto") else 0.0
        MOS1transconductance = P["kp"] if given("kp") else 0.0
        MOS1gamma = P["gamma"] if given("gamma") else 0.0
        MOS1phi = P["phi"] if given("phi") else 0.0
        MOS1substrateDoping = P["nsub"] if given("nsub") else 0.0
        MOS1tnomGiven = given("tnom")
        MOS1tnom = P["tnom"] + 273.15 if MOS1tnomGiven else 0.0
        lc_gmin = spec.gmin
        # setup (mos1.va:695-721)
        if not given("kp"):
            MOS1transconductance = 2e-5
        if not given("vto"):
            MOS1vt0 = 0
        if not given("phi"):
            MOS1phi = 0.6
        if not given("gamma"):
            MOS1gamma = 0
        # temp (mos1.va:723-897)
        if not MOS1tnomGiven:
            MOS1tnom = VACONST_tnom
        fact1 = MOS1tnom / (27.0 + 273.15)
        vtnom = MOS1tnom * CONSTKoverQ
        kt1 = 1.38064852e-23 * MOS1tnom
        egfet1 = 1.16 - 7.02e-4 * MOS1tnom * MOS1tnom / (MOS1tnom + 1108)
        arg1 = -egfet1 / (kt1 + kt1) + 1.1150877 / (1.38064852e-23 * (27.0 + 273.15 + (27.0 + 273.15)))
        pbfact1 = -2 * vtnom * (1.5 * math.log(fact1) + 1.6021766208e-19 * arg1)
        if MOS1phi <= 0.0:
            raise ValueError("Phi is not positive.")
        if (not given("tox")) or P["tox"] == 0:
            MOS1oxideCapFactor = 0
        else:
            MOS1oxideCapFactor = 3.9 * 8.854214871e-12 / P["tox"]
            if not given("kp"):
                MOS1transconductance = P["u0"] * MOS1oxideCapFactor * 1e-4
            if given("nsub"):
                if MOS1substrateDoping * 1e6 > 1.45e16:
                    if not given("phi"):
                        MOS1phi = 2 * vtnom * math.log(MOS1substrateDoping * 1e6 / 1.45e16)
                        MOS1phi = 0.1 if 0.1 > MOS1phi else MOS1phi
                    fermis = MOS1type * 0.5 * MOS1phi
                    wkfng = 3.2
                    if P["tpg"] != 0:
                        fermig = MOS1type * P["tpg"] * 0.5 * egfet1
                        wkfng = 3.25 + 0.5 * egfet1 - fermig
                    wkfngs = wkfng - (3.25 + 0.5 * egfet1 + fermis)
                    if not given("gamma"):
                        MOS1gamma = math.sqrt(2 * 11.70 * 8.854214871e-12 * 1.6021766208e-19 * MOS1substrateDoping * 1e6) / MOS1oxideCapFactor
                    if not given("vto"):
                        vfb = wkfngs - P["nss"] * 1e4 * 1.6021766208e-19 / MOS1oxideCapFactor
                        MOS1vt0 = vfb + MOS1type * (MOS1gamma * math.sqrt(MOS1phi) + MOS1phi)
                else:
                    raise ValueError("Nsub < Ni")
        if not MOS1tempGiven:
            MOS1temp = T_K + P["dtemp"]
        vt = MOS1temp * CONSTKoverQ
        ratio = MOS1temp / MOS1tnom
        fact2 = MOS1temp / (27.0 + 273.15)
        kt = MOS1temp * 1.38064852e-23
        egfet = 1.16 - 7.02e-4 * MOS1temp * MOS1temp / (MOS1temp + 1108)
        arg = -egfet / (kt + kt) + 1.1150877 / (1.38064852e-23 * (27.0 + 273.15 + (27.0 + 273.15)))
        pbfact = -2 * vt * (1.5 * math.log(fact2) + 1.6021766208e-19 * arg)
        if not given("ad"):
            MOS1drainArea = defad
        if not given("l"):
            MOS1l = defl
        if not given("as"):
            MOS1sourceArea = defas
        if not given("w"):
            MOS1w = defw
        ratio4 = ratio * math.sqrt(ratio)
        MOS1tTransconductance = MOS1transconductance / ratio4
        phio = (MOS1phi - pbfact1) / fact1
        MOS1tPhi = fact2 * phio + pbfact
        MOS1tVbi = MOS1vt0 - MOS1type * (MOS1gamma * math.sqrt(MOS1phi)) + 0.5 * (egfet1 - egfet) + MOS1type * 0.5 * (MOS1tPhi - MOS1phi)
        MOS1tVto = MOS1tVbi + MOS1type * MOS1gamma * math.sqrt(MOS1tPhi)
        MOS1tSatCur = P["is"] * math.exp(-egfet / vt + egfet1 / vtnom)
        MOS1tSatCurDens = P["js"] * math.exp(-egfet / vt + egfet1 / vtnom)
        pbo = (P["pb"] - pbfact1) / fact1
        gmaold = (P["pb"] - pbo) / pbo
        mj, mjsw, fc = P["mj"], P["mjsw"], P["fc"]
        capfact = 1 / (1 + mj * (4e-4 * (MOS1tnom - (27.0 + 273.15)) - gmaold))
        MOS1tCbd = P["cbd"] * capfact
        MOS1tCbs = P["cbs"] * capfact
        MOS1tCj = P["cj"] * capfact
        capfact = 1 / (1 + mjsw * (4e-4 * (MOS1tnom - (27.0 + 273.15)) - gmaold))
        MOS1tCjsw = P["cjsw"] * capfact
        MOS1tBulkPot = fact2 * pbo + pbfact
        gmanew = (MOS1tBulkPot - pbo) / pbo
        capfact = 1 + mj * (4e-4 * (MOS1temp - (27.0 + 273.15)) - gmanew)
        MOS1tCbd = MOS1tCbd * capfact
        MOS1tCbs = MOS1tCbs * capfact
        MOS1tCj = MOS1tCj * capfact
        capfact = 1 + mjsw * (4e-4 * (MOS1temp - (27.0 + 273.15)) - gmanew)
        MOS1tCjsw = MOS1tCjsw * capfact
        MOS1tDepCap = fc * MOS1tBulkPot
        if MOS1tSatCurDens == 0 or MOS1drainArea == 0 or MOS1sourceArea == 0:
            MOS1drainVcrit = vt * math.log(vt / (CONSTroot2 * MOS1tSatCur))
            MOS1sourceVcrit = MOS1drainVcrit
        else:
            MOS1drainVcrit = vt * math.log(vt / (CONSTroot2 * MOS1tSatCurDens * MOS1drainArea))
            MOS1sourceVcrit = vt * math.log(vt / (CONSTroot2 * MOS1tSatCurDens * MOS1sourceArea))
        if given("cbd"):
            czbd = MOS1tCbd
        elif given("cj"):
            czbd = MOS1tCj * MOS1drainArea
        else:
            czbd = 0
        czbdsw = MOS1tCjsw * MOS1drainPerimiter if given("cjsw") else 0
        arg = 1 - fc
        sarg = math.exp(-mj * math.log(arg))
        sargsw = math.exp(-mjsw * math.log(arg))
        MOS1Cbd = czbd
        MOS1Cbdsw = czbdsw
        MOS1f2d = czbd * (1 - fc * (1 + mj)) * sarg / arg + czbdsw * (1 - fc * (1 + mjsw)) * sargsw / arg
        MOS1f3d = czbd * mj * sarg / arg / MOS1tBulkPot + czbdsw * mjsw * sargsw / arg / MOS1tBulkPot
        MOS1f4d = (czbd * MOS1tBulkPot * (1 - arg * sarg) / (1 - mj) + czbdsw * MOS1tBulkPot * (1 - arg * sargsw) / (1 - mjsw)
                   - MOS1f3d / 2 * (MOS1tDepCap * MOS1tDepCap) - MOS1tDepCap * MOS1f2d)
        if given("cbs"):
            czbs = MOS1tCbs
        elif given("cj"):
            czbs = MOS1tCj * MOS1sourceArea
        else:
            czbs = 0
        czbssw = MOS1tCjsw * MOS1sourcePerimiter if given("cjsw") else 0
        MOS1Cbs = czbs
        MOS1Cbssw = czbssw
        MOS1f2s = czbs * (1 - fc * (1 + mj)) * sarg / arg + czbssw * (1 - fc * (1 + mjsw)) * sargsw / arg
        MOS1f3s = czbs * mj * sarg / arg / MOS1tBulkPot + czbssw * mjsw * sargsw / arg / MOS1tBulkPot
        MOS1f4s = (czbs * MOS1tBulkPot * (1 - arg * sarg) / (1 - mj) + czbssw * MOS1tBulkPot * (1 - arg * sargsw) / (1 - mjsw)
                   - MOS1f3s / 2 * (MOS1tDepCap * MOS1tDepCap) - MOS1tDepCap * MOS1f2s)
        rd, rs, rsh, nrd, nrs = P["rd"], P["rs"], P["rsh"], P["nrd"], P["nrs"]
        if given("rd"):
            MOS1drainConductance = 1.0 / rd if rd != 0 else 0
        elif given("rsh"):
            MOS1drainConductance = 1.0 / (rsh * nrd) if rsh != 0 else 0
        else:
            MOS1drainConductance = 0
        if given("rs"):
            MOS1sourceConductance = 1.0 / rs if rs != 0 else 0
        elif given("rsh"):
            MOS1sourceConductance = 1.0 / (rsh * nrs) if (rsh != 0 and nrs != 0) else 0
        else:
            MOS1sourceConductance = 0

        # load (mos1.va:898-1162)
        load_vt = CONSTKoverQ * MOS1temp
        EffectiveLength = MOS1l - 2 * P["ld"]
        if MOS1tSatCurDens == 0 or MOS1drainArea == 0 or MOS1sourceArea == 0:
            DrainSatCur = MOS1tSatCur
            SourceSatCur = MOS1tSatCur
        else:
            DrainSatCur = MOS1tSatCurDens * MOS1drainArea
            SourceSatCur = MOS1tSatCurDens * MOS1sourceArea
        GateSourceOverlapCap = P["cgso"] * MOS1w
        GateDrainOverlapCap = P["cgdo"] * MOS1w
        GateBulkOverlapCap = P["cgbo"] * EffectiveLength
        Beta = MOS1tTransconductance * MOS1w / EffectiveLength
        OxideCap = MOS1oxideCapFactor * EffectiveLength * MOS1w
        limited = 0
        oldget = lambda vnew, vold: vold  # DEVlimitOldGet mos1.va:484-490
        MOS1vgs = MOS1type * limit_site(1, Vb("g", "s_int"), oldget)
        MOS1vds = MOS1type * limit_site(2, Vb("d_int", "s_int"), oldget)
        MOS1vbs = MOS1type * limit_site(3, Vb("b", "s_int"), oldget)
        MOS1vbd = MOS1type * limit_site(4, Vb("b", "d_int"), oldget)
        MOS1mode = 1 if MOS1vds >= 0 else -1
        sel = MOS1vbs if MOS1mode == 1 else MOS1vbd
        if sel <= 0:
            load_sarg = dsqrt(MOS1tPhi - sel)
        else:
            load_sarg = math.sqrt(MOS1tPhi)
            # mos1.va:940 uses (MOS1mode ? vbs : vbd); MOS1mode is +-1, so always vbs
            load_sarg = load_sarg - MOS1vbs / (load_sarg + load_sarg)
            load_sarg = dmax(0, load_sarg)
        MOS1von = (MOS1tVbi * MOS1type) + MOS1gamma * load_sarg
        load_vbs = MOS1type * Vb("b", "s_int")
        load_vgs = MOS1type * Vb("g", "s_int")
        load_vds = MOS1type * Vb("d_int", "s_int")
        load_vbd = load_vbs - load_vds
        vgd = load_vgs - load_vds
        vgdo = MOS1vgs - MOS1vds
        load_von = MOS1type * MOS1von
        if MOS1vds >= 0:
            load_vgs, limited = DEVfetlim(load_vgs, MOS1vgs, load_von, limited)
            load_vds = load_vgs - vgd
            load_vds, limited = DEVlimvds(load_vds, MOS1vds, limited)
            vgd = load_vgs - load_vds
        else:
            vgd, limited = DEVfetlim(vgd, vgdo, load_von, limited)
            load_vds = load_vgs - vgd
            if not oldlimit:
                t_, limited = DEVlimvds(-load_vds, -MOS1vds, limited)
                load_vds = -t_
            load_vgs = vgd + load_vds
        if load_vds >= 0:
            load_vbs, limited = DEVpnjlim(load_vbs, MOS1vbs, load_vt, MOS1sourceVcrit, limited)
            load_vbd = load_vbs - load_vds
        else:
            load_vbd, limited = DEVpnjlim(load_vbd, MOS1vbd, load_vt, MOS1drainVcrit, limited)
            load_vbs = load_vbd + load_vds
        if initjct:  # initialize_limiting() -> $simparam("iniLim") -> ctx.initjct  (vasim.jl:1198-1206)
            load_vbs = -1
            load_vgs = MOS1type * MOS1tVto
            load_vds = 0
            load_vbd = load_vbs - load_vds

        def newset(new_value):  # DEVlimitNewSet mos1.va:492-501
            return lambda vnew, vold: new_value
        load_vgs = MOS1type * limit_site(5, Vb("g", "s_int"), newset(MOS1type * load_vgs))
        load_vds = MOS1type * limit_site(6, Vb("d_int", "s_int"), newset(MOS1type * load_vds))
        load_vbs = MOS1type * limit_site(7, Vb("b", "s_int"), newset(MOS1type * load_vbs))
        load_vbd = MOS1type * limit_site(8, Vb("b", "d_int"), newset(MOS1type * load_vbd))
        load_vbd = load_vbs - load_vds
        vgd = load_vgs - load_vds
        vgb = load_vgs - load_vbs
        if load_vbs <= -3 * load_vt:
            MOS1gbs = lc_gmin / mfactor
            MOS1cbs = lc_gmin / mfactor * load_vbs - SourceSatCur
        else:
            a_ = load_vbs / load_vt
            evbs = dexp(709.0 if 709.0 < a_ else a_)
            MOS1gbs = SourceSatCur * evbs / load_vt + lc_gmin / mfactor
            MOS1cbs = SourceSatCur * (evbs - 1) + lc_gmin / mfactor * load_vbs
        if load_vbd <= -3 * load_vt:
            MOS1gbd = lc_gmin / mfactor
            MOS1cbd = lc_gmin / mfactor * load_vbd - DrainSatCur
        else:
            a_ = load_vbd / load_vt
            evbd = dexp(709.0 if 709.0 < a_ else a_)
            MOS1gbd = DrainSatCur * evbd / load_vt + lc_gmin / mfactor
            MOS1cbd = DrainSatCur * (evbd - 1) + lc_gmin / mfactor * load_vbd
        MOS1mode = 1 if load_vds >= 0 else -1
        sel = load_vbs if MOS1mode == 1 else load_vbd
        if sel <= 0:
            load_sarg = dsqrt(MOS1tPhi - sel)
        else:
            load_sarg = math.sqrt(MOS1tPhi)
            load_sarg = load_sarg - sel / (load_sarg + load_sarg)
            load_sarg = 0 if 0 > load_sarg else load_sarg
        load_von = MOS1tVbi * MOS1type + MOS1gamma * load_sarg
        vgst = (load_vgs if MOS1mode == 1 else vgd) - load_von
        load_vdsat = vgst if vgst > 0 else 0
        if load_sarg <= 0:
            load_arg = 0
        else:
            load_arg = MOS1gamma / (load_sarg + load_sarg)
        if vgst <= 0:
            cdrain = 0
        else:
            betap = Beta * (1 + lam * (load_vds * MOS1mode))
            if vgst <= load_vds * MOS1mode:
                cdrain = betap * vgst * vgst * 0.5
            else:
                cdrain = betap * (load_vds * MOS1mode) * (vgst - 0.5 * (load_vds * MOS1mode))
        if OxideCap == 0:
            meyer_scale = 0
            meyer_unscale = 1
        else:
            meyer_scale = OxideCap
            meyer_unscale = OxideCap
        if MOS1Cbs != 0 or MOS1Cbssw != 0:
            if load_vbs < MOS1tDepCap:
                load1_arg = 1 - load_vbs / MOS1tBulkPot
                if mj == mjsw:
                    if mj == 0.5:
                        load_sargsw = 1 / dsqrt(load1_arg)
                        load1_sarg = load_sargsw
                    else:
                        load_sargsw = dexp(-mj * dln(load1_arg))
                        load1_sarg = load_sargsw
                else:
                    load1_sarg = 1 / dsqrt(load1_arg) if mj == 0.5 else dexp(-mj * dln(load1_arg))
                    load_sargsw = 1 / dsqrt(load1_arg) if mjsw == 0.5 else dexp(-mjsw * dln(load1_arg))
                MOS1qbs = MOS1tBulkPot * (MOS1Cbs * (1 - load1_arg * load1_sarg) / (1 - mj) + MOS1Cbssw * (1 - load1_arg * load_sargsw) / (1 - mjsw))
            else:
                MOS1qbs = MOS1f4s + load_vbs * (MOS1f2s + load_vbs * (MOS1f3s / 2))
        else:
            MOS1qbs = 0.0
        if MOS1Cbd != 0 or MOS1Cbdsw != 0:
            if load_vbd < MOS1tDepCap:
                load2_arg = 1 - load_vbd / MOS1tBulkPot
                if mj == 0.5 and mjsw == 0.5:
                    load_sargsw = 1 / dsqrt(load2_arg)
                    load2_sarg = load_sargsw
                else:
                    load2_sarg = 1 / dsqrt(load2_arg) if mj == 0.5 else dexp(-mj * dln(load2_arg))
                    load_sargsw = 1 / dsqrt(load2_arg) if mjsw == 0.5 else dexp(-mjsw * dln(load2_arg))
                MOS1qbd = MOS1tBulkPot * (MOS1Cbd * (1 - load2_arg * load2_sarg) / (1 - mj) + MOS1Cbdsw * (1 - load2_arg * load_sargsw) / (1 - mjsw))
            else:
                MOS1qbd = MOS1f4d + load_vbd * (MOS1f2d + load_vbd * MOS1f3d / 2)
        else:
            MOS1qbd = 0.0
        MOS1cqbd = va_ddt(MOS1qbd)
        MOS1cbd = MOS1cbd + MOS1cqbd
        MOS1cqbs = va_ddt(MOS1qbs)
        MOS1cbs = MOS1cbs + MOS1cqbs
        if MOS1mode > 0:
            MOS1capgs, MOS1capgd, MOS1capgb = DEVqmeyer(load_vgs, vgd, vgb, load_von, load_vdsat, MOS1tPhi, OxideCap)
        else:
            MOS1capgd, MOS1capgs, MOS1capgb = DEVqmeyer(vgd, load_vgs, vgb, load_von, load_vdsat, MOS1tPhi, OxideCap)
        capgs = MOS1capgs + MOS1capgs + GateSourceOverlapCap
        capgd = MOS1capgd + MOS1capgd + GateDrainOverlapCap
        capgb = MOS1capgb + MOS1capgb + GateBulkOverlapCap
        gcgs = gcgd = gcgb = 0
        ceqgs = capgs * (va_ddt(meyer_scale * load_vgs) / meyer_unscale)
        ceqgd = capgd * (va_ddt(meyer_scale * vgd) / meyer_unscale)
        ceqgb = capgb * (va_ddt(meyer_scale * vgb) / meyer_unscale)
        ceqbs = MOS1type * MOS1cbs
        ceqbd = MOS1type * MOS1cbd
        if MOS1mode >= 0:
            cdreq = MOS1type * cdrain
        else:
            cdreq = -MOS1type * cdrain
        # residuals (mos1.va:1164-1169)
        I = {}
        I["d"] = MOS1drainConductance * Vb("d", "d_int")
        I["g"] = gcgb * Vb("g", "b") + gcgd * Vb("g", "d_int") + gcgs * Vb("g", "s_int") + MOS1type * (ceqgs + ceqgb + ceqgd)
        I["s"] = MOS1sourceConductance * Vb("s", "s_int")
        I["b"] = gcgb * Vb("b", "g") + (ceqbs + ceqbd - MOS1type * ceqgb)
        I["d_int"] = MOS1drainConductance * Vb("d_int", "d") + gcgd * Vb("d_int", "g") + -(ceqbd - cdreq + MOS1type * ceqgd)
        I["s_int"] = MOS1sourceConductance * Vb("s_int", "s") + gcgs * Vb("s_int", "g") + -(cdreq + ceqbs + MOS1type * ceqgs)
        self.last_tVto = MOS1tVto
        return I


def stamp_mos1(ctx, model, d, g, s, b, x, spec, instance, mfactor=1.0):
    """The generated stamp! body for sp_mos1 (vasim.jl:3886-3963)."""
    node = {"d": d, "g": g, "s": s, "b": b}
    # internal node allocation with short-circuit aliasing (vasim.jl:3533-3564)
    node["d_int"] = d if model.sc_d() else ctx.alloc_internal_node(instance + "_sp_mos1_d_int")
    node["s_int"] = s if model.sc_s() else ctx.alloc_internal_node(instance + "_sp_mos1_s_int")
    # $limit preamble (vasim.jl:3110-3138)
    lidx, vold = [], []
    for (ps, ns) in LIMIT_BRANCHES:
        li = ctx.alloc_limit("%s_sp_mos1_lim_%s_%s" % (instance, ps, ns), node[ps], node[ns], init=0.0)
        r = ctx.resolve_index(li)
        lidx.append(li)
        vold.append(x_at(x, r))
        ctx.stamp_G(li, li, 1.0)
        ctx.stamp_G(li, node[ps], -1.0)
        ctx.stamp_G(li, node[ns], 1.0)
    # voltage extraction (vasim.jl:3584-3599)
    Vf = [x_at(x, node[nm]) for nm in ALL_NODES]
    ctx.reset_detection_counter()  # vasim.jl:3926 (per device!)
    Vd = {nm: Dual.seed(Vf[k], k, WIDTH) for k, nm in enumerate(ALL_NODES)}  # vasim.jl:3617-3626
    limw = [0.0] * len(LIMIT_SITES)

    def limit_site(j, vnew, fn):  # vasim.jl:1258-1330 ; j is 1-based site index
        bidx = LIMIT_BRANCHES.index(LIMIT_SITES[j - 1])
        w = val(fn(vnew, vold[bidx]))
        limw[j - 1] = w
        ctx.record_limit_w(lidx[bidx], w)
        seed = np.zeros(WIDTH)
        seed[N_NODES + j - 1] = 1.0
        return vnew - val(vnew) + w + Dual(0.0, seed)

    # conditional V(d_int,d) <+ 0 / V(s_int,s) <+ 0 (vasim.jl:2313-2395): nothing when aliased
    for (pi, ni) in (("d_int", "d"), ("s_int", "s")):
        pn, nn = node[pi], node[ni]
        cond = model.sc_d() if pi == "d_int" else model.sc_s()
        if cond and pn != nn:
            Iv = ctx.alloc_current("%s_I_V_%s_%s" % (instance, pi, ni))
            ctx.stamp_G(pn, Iv, 1.0)
            ctx.stamp_G(nn, Iv, -1.0)
            ctx.stamp_G(Iv, pn, 1.0)
            ctx.stamp_G(Iv, nn, -1.0)
            ctx.stamp_b(Iv, 0.0)

    Ibr = model.evaluate(Vd, limit_site, spec, ctx.initjct, mfactor)

    def lim_delta(j):  # limit_rhs_terms vasim.jl:2957-2966
        ps, ns = LIMIT_SITES[j]
        return Vf[ALL_NODES.index(ps)] - Vf[ALL_NODES.index(ns)] - limw[j]

    for br in BRANCHES:  # vasim.jl:3284-3521 ; every branch is (br, ground)
        p_node = node[br]
        I_branch = mfactor * Ibr[br]
        if isinstance(I_branch, CDual):
            I_resist, I_react = I_branch.r, I_branch.q
            has_reactive = True
        else:
            I_resist, I_react = I_branch, 0.0
            has_reactive = False
        I_val = val(I_resist)
        dI = partials(I_resist, WIDTH)
        q_val = val(I_react)
        dq = partials(I_react, WIDTH)
        for k in range(N_NODES):
            k_node = node[ALL_NODES[k]]
            if p_node != 0 and k_node != 0:
                ctx.stamp_G(p_node, k_node, dI[k])
        if has_reactive:
            V_branch = Vf[ALL_NODES.index(br)]
            if ctx.detect_or_cached("%s_sp_mos1_Q_%s_0" % (instance, br), V_branch, q_val):
                qi = ctx.alloc_charge("%s_sp_mos1_Q_%s_0" % (instance, br), p_node, 0)
                if p_node != 0:
                    ctx.stamp_C(p_node, qi, 1.0 / CHARGE_SCALE)
                ctx.stamp_G(qi, qi, 1.0)
                for k in range(N_NODES):
                    k_node = node[ALL_NODES[k]]
                    if k_node != 0:
                        ctx.stamp_G(qi, k_node, -CHARGE_SCALE * dq[k])
                b_con = q_val
                for k in range(N_NODES):
                    b_con -= dq[k] * Vf[k]
                for j in range(len(LIMIT_SITES)):
                    b_con += dq[N_NODES + j] * lim_delta(j)
                ctx.stamp_b(qi, CHARGE_SCALE * b_con)
            else:
                for k in range(N_NODES):
                    k_node = node[ALL_NODES[k]]
                    if p_node != 0 and k_node != 0:
                        ctx.stamp_C(p_node, k_node, dq[k])
        Ieq = I_val
        for k in range(N_NODES):
            Ieq = Ieq + (-dI[k] * Vf[k])
        for j in range(len(LIMIT_SITES)):
            Ieq = Ieq + dI[N_NODES + j] * lim_delta(j)
        if p_node != 0:
            ctx.stamp_b(p_node, -Ieq)
