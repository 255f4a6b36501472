"""The oracle's DC fallback chain (oracle/mna_ref.py: dc_solve_with_fallbacks, restating solve.jl:871-929) with every
Newton run logged, in the format of the product's cadnip_dc_log: (stage, rung value, converged, Newton solves);
stage 0 = PCNR, 1 = plain Newton, 2 = gshunt stepping (value = gshunt), 3 = source stepping (value = srcFact)."""
import numpy as np

from oracle import mna_ref as M


def oracle_chain(cs, ws, u0, abstol=1e-10, maxiters=100, use_stepping=True):
    log = []
    stage = [1]
    plain, pcnr = M.dc_newton_plain, M.dc_pcnr_newton

    def plain_logged(cs_, ws_, u, abstol_=1e-10, maxiters_=100):
        r = plain(cs_, ws_, u, abstol_, maxiters_)
        value = {1: 0.0, 2: cs_.spec.gshunt, 3: cs_.spec.srcFact}[stage[0]]
        log.append((stage[0], float(value), bool(r[1]), int(r[2])))
        return r

    def pcnr_logged(cs_, ws_, u, abstol_=1e-10, maxiters_=100):
        r = pcnr(cs_, ws_, u, abstol_, maxiters_)
        log.append((0, 0.0, bool(r[1]), int(r[2])))
        return r

    def gshunt_logged(*a, **k):
        stage[0] = 2
        try:
            return gshunt(*a, **k)
        finally:
            stage[0] = 1

    def source_logged(*a, **k):
        stage[0] = 3
        try:
            return source(*a, **k)
        finally:
            stage[0] = 1

    gshunt, source = M.gshunt_stepping, M.source_stepping
    M.dc_newton_plain, M.dc_pcnr_newton, M.gshunt_stepping, M.source_stepping = plain_logged, pcnr_logged, gshunt_logged, source_logged
    try:
        with np.errstate(all="ignore"):
            u, ok = M.dc_solve_with_fallbacks(cs, ws, np.array(u0, dtype=float), abstol, maxiters, use_stepping)
    finally:
        M.dc_newton_plain, M.dc_pcnr_newton, M.gshunt_stepping, M.source_stepping = plain, pcnr, gshunt, source
    return u, ok, log


def same_ladder(got, ref):
    """Product log of one instance vs the oracle's: same sequence of (stage, rung, converged, Newton solves).  The product
    labels the first run 0 also when it is a plain Newton (circuit without limit variables); the oracle calls that 1."""
    norm = lambda L: [(1 if s == 0 and k == 0 and not any_pcnr else s, v, ok, it) for k, (s, v, ok, it) in enumerate(L)]
    any_pcnr = any(s == 0 for s, *_ in ref)
    g, r = norm(got), norm(ref)
    if len(g) != len(r):
        return False
    for (s1, v1, o1, i1), (s2, v2, o2, i2) in zip(g, r):
        if s1 != s2 or o1 != o2 or i1 != i2 or abs(v1 - v2) > 1e-12 * max(abs(v2), 1e-300):
            return False
    return True
