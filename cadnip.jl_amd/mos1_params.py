"""sp_mos1 bias-independent preprocessing, hoisted out of the per-iteration stamp.

The reference re-evaluates the model's ``setup`` and ``temp`` sections
(/root/reference/models/VADistillerModels.jl/va/mos1.va:695-897) on every stamp call
because Cadnip lowers the whole analog block into one ``stamp!`` body
(/root/reference/src/vasim.jl:3886-3963).  They depend only on (model card, instance
parameters, temperature), so the GPU path evaluates them once per (device, sweep instance)
on the host -- vectorised over the sweep instances with numpy -- and hands the kernel the
CADNIP_MOS1_NPAR derived values below (include/cadnip_hip.h).
"""
import numpy as np

NPAR = 36
(P_TYPE, P_VT, P_TPHI, P_TVBI, P_TVTO, P_GAMMA, P_LAMBDA, P_BETA, P_OXCAP, P_SSATCUR, P_DSATCUR,
 P_SVCRIT, P_DVCRIT, P_CBS, P_CBSSW, P_CBD, P_CBDSW, P_TBULKPOT, P_TDEPCAP, P_F2S, P_F3S, P_F4S,
 P_F2D, P_F3D, P_F4D, P_MJ, P_MJSW, P_CGSOV, P_CGDOV, P_CGBOV, P_GD, P_GS, P_MFACTOR, P_GMIN,
 P_RSV0, P_RSV1) = range(NPAR)

DEFAULTS = dict(
    l=0.0, w=0.0, ad=0.0, pd=0.0, ps=0.0, nrd=1.0, nrs=1.0, temp=0.0, dtemp=0.0,
    type=1, vto=0.0, kp=2e-5, gamma=0.0, phi=0.6, rd=0.0, rs=0.0, cbd=0.0, cbs=0.0,
    pb=0.8, cgso=0.0, cgdo=0.0, cgbo=0.0, rsh=0.0, cj=0.0, mj=0.5, cjsw=0.0, mjsw=0.5, js=0.0, tox=0.0,
    ld=0.0, u0=600.0, fc=0.5, nsub=0.0, tpg=1, nss=0.0, tnom=0.0)
DEFAULTS["as"] = 0.0
DEFAULTS["lambda"] = 0.0
DEFAULTS["is"] = 1e-14

KOVERQ = 1.38064852e-23 / 1.6021766208e-19
BOLTZ = 1.38064852e-23
CHARGE = 1.6021766208e-19
REFT = 27.0 + 273.15


def _uniform(x):
    """Value-dependent branches of the setup code must take the same path for every sweep
    instance (otherwise the instances do not share a structure)."""
    a = np.asarray(x, dtype=float)
    if a.ndim and not np.all(a == a.flat[0]):
        raise ValueError("sweep instances disagree on a structure-selecting sp_mos1 parameter")
    return float(a.flat[0]) if a.ndim else float(a)


def short_circuits(given):
    """(d_int aliased to d, s_int aliased to s): mos1.va:716-721, vasim.jl:3533-3564."""
    g = lambda k: _uniform(given.get(k, DEFAULTS[k]))
    sc_d = not (g("rd") != 0 or (g("rsh") != 0 and g("nrd") != 0))
    sc_s = not (g("rs") != 0 or (g("rsh") != 0 and g("nrs") != 0))
    return sc_d, sc_s


def derive(given, temp_c, tnom_c, gmin, mfactor=1.0):
    """``given``: dict of explicitly given parameters (numbers or [B] arrays).
    ``temp_c``: circuit temperature in Celsius ([B] array or scalar).
    Returns ([NPAR, B] array, info dict with structure-relevant facts)."""
    given = dict(given)
    temp_c = np.atleast_1d(np.asarray(temp_c, dtype=float))
    B = temp_c.shape[0]
    for v in given.values():
        if np.ndim(v):
            B = max(B, np.shape(v)[0])
    if temp_c.shape[0] == 1 and B > 1:
        temp_c = np.full(B, temp_c[0])

    def P(k):
        return np.asarray(given.get(k, DEFAULTS[k]), dtype=float) * np.ones(B)

    has = lambda k: k in given
    typ = int(_uniform(given.get("type", 1)))
    # field assignments + setup (mos1.va:640-721)
    l = P("l") if has("l") else np.full(B, 1e-4)       # defl
    w = P("w") if has("w") else np.full(B, 1e-4)       # defw
    ad = P("ad") if has("ad") else np.zeros(B)         # defad
    as_ = P("as") if has("as") else np.zeros(B)        # defas
    pd = P("pd") if has("pd") else np.zeros(B)
    ps = P("ps") if has("ps") else np.zeros(B)
    vt0 = P("vto") if has("vto") else np.zeros(B)
    kp = P("kp") if has("kp") else np.full(B, 2e-5)
    gamma = P("gamma") if has("gamma") else np.zeros(B)
    phi = P("phi") if has("phi") else np.full(B, 0.6)
    nsub = P("nsub") if has("nsub") else np.zeros(B)
    tnom = P("tnom") + 273.15 if has("tnom") else np.full(B, tnom_c + 273.15)
    # temp (mos1.va:723-897)
    fact1 = tnom / REFT
    vtnom = tnom * KOVERQ
    kt1 = BOLTZ * tnom
    egfet1 = 1.16 - 7.02e-4 * tnom * tnom / (tnom + 1108)
    arg1 = -egfet1 / (kt1 + kt1) + 1.1150877 / (BOLTZ * (REFT + REFT))
    pbfact1 = -2 * vtnom * (1.5 * np.log(fact1) + CHARGE * arg1)
    if np.any(phi <= 0):
        raise ValueError("Phi is not positive.")
    tox = P("tox")
    if (not has("tox")) or _uniform(tox) == 0:
        oxcapf = np.zeros(B)
    else:
        oxcapf = 3.9 * 8.854214871e-12 / tox
        if not has("kp"):
            kp = P("u0") * oxcapf * 1e-4
        if has("nsub"):
            if np.all(nsub * 1e6 > 1.45e16):
                if not has("phi"):
                    phi = 2 * vtnom * np.log(nsub * 1e6 / 1.45e16)
                    phi = np.where(0.1 > phi, 0.1, phi)
                fermis = typ * 0.5 * phi
                wkfng = np.full(B, 3.2)
                tpg = int(_uniform(given.get("tpg", 1)))
                if tpg != 0:
                    fermig = typ * tpg * 0.5 * egfet1
                    wkfng = 3.25 + 0.5 * egfet1 - fermig
                wkfngs = wkfng - (3.25 + 0.5 * egfet1 + fermis)
                if not has("gamma"):
                    gamma = np.sqrt(2 * 11.70 * 8.854214871e-12 * CHARGE * nsub * 1e6) / oxcapf
                if not has("vto"):
                    vfb = wkfngs - P("nss") * 1e4 * CHARGE / oxcapf
                    vt0 = vfb + typ * (gamma * np.sqrt(phi) + phi)
            else:
                raise ValueError("Nsub < Ni")
    T = P("temp") + 273.15 if has("temp") else (temp_c + 273.15) + P("dtemp")
    vt = T * KOVERQ
    ratio = T / tnom
    fact2 = T / REFT
    kt = T * BOLTZ
    egfet = 1.16 - 7.02e-4 * T * T / (T + 1108)
    arg = -egfet / (kt + kt) + 1.1150877 / (BOLTZ * (REFT + REFT))
    pbfact = -2 * vt * (1.5 * np.log(fact2) + CHARGE * arg)
    ratio4 = ratio * np.sqrt(ratio)
    tKp = kp / ratio4
    phio = (phi - pbfact1) / fact1
    tPhi = fact2 * phio + pbfact
    tVbi = vt0 - typ * (gamma * np.sqrt(phi)) + 0.5 * (egfet1 - egfet) + typ * 0.5 * (tPhi - phi)
    tVto = tVbi + typ * gamma * np.sqrt(tPhi)
    tSatCur = P("is") * np.exp(-egfet / vt + egfet1 / vtnom)
    tSatCurDens = P("js") * np.exp(-egfet / vt + egfet1 / vtnom)
    pb = P("pb")
    pbo = (pb - pbfact1) / fact1
    gmaold = (pb - pbo) / pbo
    mj, mjsw, fc = P("mj"), P("mjsw"), P("fc")
    capfact = 1 / (1 + mj * (4e-4 * (tnom - REFT) - gmaold))
    tCbd = P("cbd") * capfact
    tCbs = P("cbs") * capfact
    tCj = P("cj") * capfact
    capfact = 1 / (1 + mjsw * (4e-4 * (tnom - REFT) - gmaold))
    tCjsw = P("cjsw") * capfact
    tBulkPot = fact2 * pbo + pbfact
    gmanew = (tBulkPot - pbo) / pbo
    capfact = 1 + mj * (4e-4 * (T - REFT) - gmanew)
    tCbd = tCbd * capfact
    tCbs = tCbs * capfact
    tCj = tCj * capfact
    capfact = 1 + mjsw * (4e-4 * (T - REFT) - gmanew)
    tCjsw = tCjsw * capfact
    tDepCap = fc * tBulkPot
    use_is = (_uniform(np.where(tSatCurDens == 0, 0.0, 1.0)) == 0) or _uniform(ad) == 0 or _uniform(as_) == 0
    root2 = np.sqrt(2.0)
    if use_is:
        dvcrit = vt * np.log(vt / (root2 * tSatCur))
        svcrit = dvcrit
        dsat = tSatCur
        ssat = tSatCur
    else:
        dvcrit = vt * np.log(vt / (root2 * tSatCurDens * ad))
        svcrit = vt * np.log(vt / (root2 * tSatCurDens * as_))
        dsat = tSatCurDens * ad
        ssat = tSatCurDens * as_
    czbd = tCbd if has("cbd") else (tCj * ad if has("cj") else np.zeros(B))
    czbdsw = tCjsw * pd if has("cjsw") else np.zeros(B)
    a1 = 1 - fc
    sarg = np.exp(-mj * np.log(a1))
    sargsw = np.exp(-mjsw * np.log(a1))
    f2d = czbd * (1 - fc * (1 + mj)) * sarg / a1 + czbdsw * (1 - fc * (1 + mjsw)) * sargsw / a1
    f3d = czbd * mj * sarg / a1 / tBulkPot + czbdsw * mjsw * sargsw / a1 / tBulkPot
    f4d = (czbd * tBulkPot * (1 - a1 * sarg) / (1 - mj) + czbdsw * tBulkPot * (1 - a1 * sargsw) / (1 - mjsw)
           - f3d / 2 * (tDepCap * tDepCap) - tDepCap * f2d)
    czbs = tCbs if has("cbs") else (tCj * as_ if has("cj") else np.zeros(B))
    czbssw = tCjsw * ps if has("cjsw") else np.zeros(B)
    f2s = czbs * (1 - fc * (1 + mj)) * sarg / a1 + czbssw * (1 - fc * (1 + mjsw)) * sargsw / a1
    f3s = czbs * mj * sarg / a1 / tBulkPot + czbssw * mjsw * sargsw / a1 / tBulkPot
    f4s = (czbs * tBulkPot * (1 - a1 * sarg) / (1 - mj) + czbssw * tBulkPot * (1 - a1 * sargsw) / (1 - mjsw)
           - f3s / 2 * (tDepCap * tDepCap) - tDepCap * f2s)
    rd, rs, rsh, nrd, nrs = P("rd"), P("rs"), P("rsh"), P("nrd"), P("nrs")
    with np.errstate(divide="ignore", invalid="ignore"):
        if has("rd"):
            gd = np.where(rd != 0, 1.0 / rd, 0.0)
        elif has("rsh"):
            gd = np.where(rsh != 0, 1.0 / (rsh * nrd), 0.0)
        else:
            gd = np.zeros(B)
        if has("rs"):
            gs = np.where(rs != 0, 1.0 / rs, 0.0)
        elif has("rsh"):
            gs = np.where((rsh != 0) & (nrs != 0), 1.0 / (rsh * nrs), 0.0)
        else:
            gs = np.zeros(B)
    leff = l - 2 * P("ld")
    out = np.zeros((NPAR, B))
    out[P_TYPE] = typ
    out[P_VT] = vt
    out[P_TPHI] = tPhi
    out[P_TVBI] = tVbi
    out[P_TVTO] = tVto
    out[P_GAMMA] = gamma
    out[P_LAMBDA] = P("lambda")
    out[P_BETA] = tKp * w / leff
    out[P_OXCAP] = oxcapf * leff * w
    out[P_SSATCUR] = ssat
    out[P_DSATCUR] = dsat
    out[P_SVCRIT] = svcrit
    out[P_DVCRIT] = dvcrit
    out[P_CBS] = czbs
    out[P_CBSSW] = czbssw
    out[P_CBD] = czbd
    out[P_CBDSW] = czbdsw
    out[P_TBULKPOT] = tBulkPot
    out[P_TDEPCAP] = tDepCap
    out[P_F2S], out[P_F3S], out[P_F4S] = f2s, f3s, f4s
    out[P_F2D], out[P_F3D], out[P_F4D] = f2d, f3d, f4d
    out[P_MJ], out[P_MJSW] = mj, mjsw
    out[P_CGSOV] = P("cgso") * w
    out[P_CGDOV] = P("cgdo") * w
    out[P_CGBOV] = P("cgbo") * leff
    out[P_GD], out[P_GS] = gd, gs
    out[P_MFACTOR] = mfactor
    out[P_GMIN] = gmin
    info = {
        "oxcap_nonzero": bool(np.any(out[P_OXCAP] != 0)),
        "cbs_nonzero": bool(np.any(czbs != 0) or np.any(czbssw != 0)),
        "cbd_nonzero": bool(np.any(czbd != 0) or np.any(czbdsw != 0)),
    }
    return out, info


# ---------------------------------------------------------------------------------------------
# value-only evaluation of the four reactive branch charges, used once on the host by the
# structure-discovery detection passes (structure.detect_mos1_vdep).  Mirrors the load section
# mos1.va:898-1162 (limiting at mos1.va:919-980, charges :1049-1147) without derivatives.
# ---------------------------------------------------------------------------------------------
def _fetlim(vnew, vold, vto):
    vlim = vnew
    vtsthi = abs(2 * (vold - vto)) + 2
    vtstlo = abs(vold - vto) + 1
    vtox = vto + 3.5
    delv = vnew - vold
    if vold >= vto:
        if vold >= vtox:
            if delv <= 0:
                if vlim >= vtox:
                    if -delv > vtstlo:
                        vlim = vold - vtstlo
                else:
                    vlim = max(vnew, vto + 2)
            elif delv >= vtsthi:
                vlim = vold + vtsthi
        else:
            vlim = max(vnew, vto - 0.5) if delv <= 0 else min(vnew, vto + 4)
    else:
        if delv <= 0:
            if -delv > vtsthi:
                vlim = vold - vtsthi
        else:
            vtemp = vto + 0.5
            if vnew <= vtemp:
                if delv > vtstlo:
                    vlim = vold + vtstlo
            else:
                vlim = vtemp
    return vlim


def _limvds(vnew, vold):
    if vold >= 3.5:
        if vnew > vold:
            return min(vnew, 3 * vold + 2)
        return max(vnew, 2.0) if vnew < 3.5 else vnew
    return min(vnew, 4.0) if vnew > vold else max(vnew, -0.5)


def _pnjlim(vnew, vold, vt, vcrit):
    import math
    if vnew > vcrit and abs(vnew - vold) > vt + vt:
        if vold > 0:
            arg = (vnew - vold) / vt
            return vold + vt * math.log(1 + arg) if arg > 0 else vold - vt * math.log(1 - arg)
        return vt * math.log(vnew / vt)
    if vnew < 0:
        arg = -vold - 1 if vold > 0 else 2 * vold - 1
        if vnew < arg:
            return arg
    return vnew


def _qmeyer(vgs, vgd, von, vdsat, phi, cox):
    vgst = vgs - von
    vdsat = vdsat if vdsat > 0.025 else 0.025
    if vgst <= -phi:
        return 0.0, 0.0, cox / 2
    if vgst <= -phi / 2:
        return 0.0, 0.0, -vgst * cox / (2 * phi)
    if vgst <= 0:
        capgb = -vgst * cox / (2 * phi)
        capgs = vgst * cox / (1.5 * phi) + cox / 3
        vds = vgs - vgd
        if vds >= vdsat:
            return capgs, 0.0, capgb
        vddif = 2.0 * vdsat - vds
        vddif1 = vdsat - vds
        vddif2 = vddif * vddif
        return capgs * (1.0 - vddif1 * vddif1 / vddif2), capgs * (1.0 - vdsat * vdsat / vddif2), capgb
    vds = vgs - vgd
    if vdsat <= vds:
        return cox / 3, 0.0, 0.0
    vddif = 2.0 * vdsat - vds
    vddif1 = vdsat - vds
    vddif2 = vddif * vddif
    return cox * (1.0 - vddif1 * vddif1 / vddif2) / 3, cox * (1.0 - vdsat * vdsat / vddif2) / 3, 0.0


def _qdep(v, Cb, Cbsw, pot, depcap, mj, mjsw, f2, f3, f4):
    import math
    if Cb != 0 or Cbsw != 0:
        if v < depcap:
            arg = 1 - v / pot
            sarg = 1 / math.sqrt(arg) if mj == 0.5 else math.exp(-mj * math.log(arg))
            sargsw = 1 / math.sqrt(arg) if mjsw == 0.5 else math.exp(-mjsw * math.log(arg))
            return pot * (Cb * (1 - arg * sarg) / (1 - mj) + Cbsw * (1 - arg * sargsw) / (1 - mjsw))
        return f4 + v * (f2 + v * (f3 / 2))
    return 0.0


def host_charges(P, V6, vold4):
    """P: [NPAR] derived parameters of one device; V6 = (Vd,Vg,Vs,Vb,Vdi,Vsi); vold4: the four limit
    unknowns.  Returns (q_g, q_b, q_dint, q_sint), each already scaled by $mfactor."""
    import math
    typ, vt, tPhi, tVbi, gamma = P[P_TYPE], P[P_VT], P[P_TPHI], P[P_TVBI], P[P_GAMMA]
    ox, mf = P[P_OXCAP], P[P_MFACTOR]
    Vd, Vg, Vs, Vb, Vdi, Vsi = V6
    o_vgs, o_vds, o_vbs, o_vbd = (typ * v for v in vold4)
    osel = o_vbs if o_vds >= 0 else o_vbd
    if osel <= 0:
        osarg = math.sqrt(max(tPhi - osel, 0.0))
    else:
        osarg = math.sqrt(tPhi)
        osarg = max(0.0, osarg - o_vbs / (osarg + osarg))
    von = typ * ((tVbi * typ) + gamma * osarg)
    vbs, vgs, vds = typ * (Vb - Vsi), typ * (Vg - Vsi), typ * (Vdi - Vsi)
    vbd, vgd, vgdo = vbs - vds, vgs - vds, o_vgs - o_vds
    if o_vds >= 0:
        vgs = _fetlim(vgs, o_vgs, von)
        vds = _limvds(vgs - vgd, o_vds)
        vgd = vgs - vds
    else:
        vgd = _fetlim(vgd, vgdo, von)
        vds = -_limvds(-(vgs - vgd), -o_vds)
        vgs = vgd + vds
    if vds >= 0:
        vbs = _pnjlim(vbs, o_vbs, vt, P[P_SVCRIT])
        vbd = vbs - vds
    else:
        vbd = _pnjlim(vbd, o_vbd, vt, P[P_DVCRIT])
        vbs = vbd + vds
    vbd = vbs - vds
    vgd = vgs - vds
    vgb = vgs - vbs
    mode = 1 if vds >= 0 else -1
    sel = vbs if mode == 1 else vbd
    if sel <= 0:
        sarg = math.sqrt(max(tPhi - sel, 0.0))
    else:
        s0 = math.sqrt(tPhi)
        sarg = max(0.0, s0 - sel / (s0 + s0))
    lvon = tVbi * typ + gamma * sarg
    vgst = (vgs if mode == 1 else vgd) - lvon
    vdsat = vgst if vgst > 0 else 0.0
    qbs = _qdep(vbs, P[P_CBS], P[P_CBSSW], P[P_TBULKPOT], P[P_TDEPCAP], P[P_MJ], P[P_MJSW], P[P_F2S], P[P_F3S], P[P_F4S])
    qbd = _qdep(vbd, P[P_CBD], P[P_CBDSW], P[P_TBULKPOT], P[P_TDEPCAP], P[P_MJ], P[P_MJSW], P[P_F2D], P[P_F3D], P[P_F4D])
    if mode > 0:
        cgs, cgd, cgb = _qmeyer(vgs, vgd, lvon, vdsat, tPhi, ox)
    else:
        cgd, cgs, cgb = _qmeyer(vgd, vgs, lvon, vdsat, tPhi, ox)
    capgs, capgd, capgb = 2 * cgs + P[P_CGSOV], 2 * cgd + P[P_CGDOV], 2 * cgb + P[P_CGBOV]
    ms, mu = (0.0, 1.0) if ox == 0 else (ox, ox)
    qgs, qgd, qgb = capgs * (ms * vgs / mu), capgd * (ms * vgd / mu), capgb * (ms * vgb / mu)
    return (mf * typ * (qgs + qgb + qgd), mf * (typ * qbs + typ * qbd - typ * qgb),
            mf * -(typ * qbd + typ * qgd), mf * -(typ * qbs + typ * qgs))
