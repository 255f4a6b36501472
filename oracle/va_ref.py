"""TEST INFRASTRUCTURE (oracle) -- the generated stamp! of an arbitrary Verilog-A module, interpreted.

Restates /root/reference/src/vasim.jl:2993-3985 (generate_mna_stamp_method_nterm) for the module subset the product's
front end parses: node voltages are seeded as ``Dual{JacobianTag}`` (one partial per module node, :3617-3626), the
analog block is evaluated on them with ``ddt`` producing ``Dual{ContributionTag}`` pairs (contrib.jl:356-375), and every
branch (p, n) is stamped as in :3319-3521 -- G[p,k] += dI/dV_k, G[n,k] -= ..., equivalent currents into b, and the
reactive part either through a charge unknown (:3433-3472) or as constant capacitances (:3474-3482), chosen by
``ctx.detect_or_cached`` (contrib.jl:214-257).  ``$limit`` sites follow vasim.jl:1258-1330 (site dual anchored at the
limiter's value, own partial slot), the hoisted limit preamble :3110-3138 and the lim_rhs terms :2957-2966.

The syntax tree comes from the product's parser (cadnip.jl_amd/va/frontend.py); the arithmetic, the dual numbers
(oracle/dual.py) and the stamping are the oracle's own.  Pure-Python: meant for small cases.
"""
import math

import numpy as np

from .dual import CDual, Dual, dabs, dexp, dln, dmax, dmin, dsqrt, partials, va_ddt, val
from .mna_ref import x_at

CHARGE_SCALE = 1e12            # contrib.jl:39
K_BOLTZ, Q_ELEM = 1.380649e-23, 1.602176634e-19


def _chain(x, f, df):
    if isinstance(x, Dual):
        return Dual(f, x.p * df)
    return f


def _unary(fn, dfn):
    def g(x):
        v = val(x)
        f = fn(v)
        return _chain(x, f, dfn(v, f))
    return g


def _pow(a, b):
    av, bv = val(a), val(b)
    f = math.pow(av, bv)
    p = 0.0
    if isinstance(a, Dual):
        p = p + a.p * (0.0 if bv == 0.0 else bv * math.pow(av, bv - 1.0))
    if isinstance(b, Dual) and np.any(b.p != 0.0):
        p = p + b.p * (f * math.log(av))
    if isinstance(a, Dual) or isinstance(b, Dual):
        return Dual(f, p + np.zeros(len(a.p if isinstance(a, Dual) else b.p)))
    return f


def _limexp(v):
    return math.exp(v) if v < 80.0 else math.exp(80.0) * (1.0 + v - 80.0)


FUNCS = {
    "exp": dexp, "ln": dln, "sqrt": dsqrt, "abs": dabs, "pow": _pow,
    "log": _unary(math.log10, lambda v, f: 1.0 / (v * math.log(10.0))),
    "limexp": _unary(_limexp, lambda v, f: f if v < 80.0 else math.exp(80.0)),
    "tanh": _unary(math.tanh, lambda v, f: 1.0 - f * f), "sinh": _unary(math.sinh, lambda v, f: math.cosh(v)),
    "cosh": _unary(math.cosh, lambda v, f: math.sinh(v)), "sin": _unary(math.sin, lambda v, f: math.cos(v)),
    "cos": _unary(math.cos, lambda v, f: -math.sin(v)), "atan": _unary(math.atan, lambda v, f: 1.0 / (1.0 + v * v)),
    # ties take the first operand (the product's va_max / va_min)
    "max": lambda a, b: b if val(b) > val(a) else a, "min": lambda a, b: b if val(b) < val(a) else a,
}


def evaluate(mod, Vd, par, temp_k, mfactor, gmin, limit_site=None, initjct=False):
    """Branch contributions of ``mod`` on dual node voltages ``Vd``: one Dual / CDual / float per branch.
    ``limit_site(j, vnew_dual, fn)`` implements a $limit call site (stamp_va)."""
    env = {v: 0.0 for v in mod.locals_}
    acc = [0.0 for _ in mod.branches]
    scope = [None]

    def call(fname, args):
        f_in, f_loc, f_body = mod.functions[fname]
        saved = scope[0]
        fenv = dict({v: 0.0 for v in f_loc}, **dict(zip(f_in, args)), **{fname: 0.0})
        scope[0] = fenv
        run(f_body)
        scope[0] = saved
        return fenv[fname]

    def ev(e):
        k = e[0]
        if k == "num":
            return e[1]
        if k == "var":
            if scope[0] is not None:
                return scope[0][e[1]]
            return par[e[1]] if e[1] in par else env[e[1]]
        if k == "ucall":
            return call(e[1], [ev(a) for a in e[2]])
        if k == "limit":
            a, b = mod.node_index(e[1]), mod.node_index(e[2])
            vnew = (Vd[a] if a >= 0 else 0.0) - (Vd[b] if b >= 0 else 0.0)
            user = [ev(x) for x in e[4]]
            return limit_site(e[5][0], vnew, lambda vn, vo: call(e[3], [vn, vo] + user))
        if k == "V":
            a, b = mod.node_index(e[1]), mod.node_index(e[2])
            return (Vd[a] if a >= 0 else 0.0) - (Vd[b] if b >= 0 else 0.0)
        if k == "ddt":
            return va_ddt(ev(e[1]))
        if k == "un":
            x = ev(e[2])
            return (0.0 if val(x) else 1.0) if e[1] == "!" else -x
        if k == "cond":
            return ev(e[2]) if val(ev(e[1])) else ev(e[3])
        if k == "call":
            return FUNCS[e[1]](*[ev(a) for a in e[2]])
        if k == "sys":
            if e[1] == "$temperature":
                return temp_k
            if e[1] == "$vt":
                return K_BOLTZ * (ev(e[2][0]) if e[2] else temp_k) / Q_ELEM
            if e[1] == "$mfactor":
                return mfactor
            if e[2] and e[2][0] == ("str", "gmin"):
                return gmin
            if e[2] and e[2][0] == ("str", "initjct"):
                return 1.0 if initjct else 0.0
            return ev(e[2][1])
        op, l, r = e[1], ev(e[2]), ev(e[3])
        if op == "+":
            return l + r
        if op == "-":
            return l - r
        if op == "*":
            return l * r
        if op == "/":
            return l / r
        lv, rv = val(l), val(r)
        return float({"==": lv == rv, "!=": lv != rv, "<": lv < rv, ">": lv > rv, "<=": lv <= rv, ">=": lv >= rv,
                      "&&": bool(lv) and bool(rv), "||": bool(lv) or bool(rv)}[op])

    def run(stmts):
        for s in stmts:
            if s[0] == "assign":
                (scope[0] if scope[0] is not None else env)[s[1]] = ev(s[2])
            elif s[0] == "contrib":
                b = mod.branches.index((mod.node_index(s[1]), mod.node_index(s[2])))
                acc[b] = acc[b] + ev(s[3])
            elif s[0] == "block":
                run(s[1])
            elif s[0] == "if":
                run([s[2]] if val(ev(s[1])) else [s[3]])
            # "short" (V(a,b) <+ 0): structural, nothing to evaluate

    run(mod.body)
    return acc


def stamp_va(ctx, mod, ext_nodes, x, par, spec, instance, mfactor=1.0, gmin=None):
    """The generated stamp! body for one instance of ``mod`` (vasim.jl:3886-3963)."""
    N, S = len(mod.nodes), mod.n_sites
    W = N + S
    # internal node allocation with short-circuit aliasing (vasim.jl:3533-3564): V(a,b) <+ 0 under a parameter condition
    alias = mod.aliases(par)
    node = list(ext_nodes) + [None] * (N - len(mod.ports))
    for k in range(len(mod.ports), N):
        if k not in alias:
            node[k] = ctx.alloc_internal_node("%s_%s_%s" % (instance, mod.name, mod.nodes[k]))
    for k in range(len(mod.ports), N):
        if k in alias:
            node[k] = node[alias[k]]
    # $limit preamble (vasim.jl:3110-3138): one limit unknown per probe branch, its tracking row u_l - (V_p - V_n) = 0
    lidx, vold = [], []
    for (pl, nl) in mod.limit_branches:
        p_node, n_node = (node[pl] if pl >= 0 else 0), (node[nl] if nl >= 0 else 0)
        li = ctx.alloc_limit("%s_%s_lim_%s_%s" % (instance, mod.name, mod.nodes[pl] if pl >= 0 else "0", mod.nodes[nl] if nl >= 0 else "0"),
                             p_node, n_node, init=0.0)
        lidx.append(li)
        vold.append(x_at(x, ctx.resolve_index(li)))
        ctx.stamp_G(li, li, 1.0)
        ctx.stamp_G(li, p_node, -1.0)
        ctx.stamp_G(li, n_node, 1.0)
    Vf = [x_at(x, nd) for nd in node]
    ctx.reset_detection_counter()                                                  # vasim.jl:3926
    Vd = [Dual.seed(Vf[k], k, W) for k in range(N)]                                # vasim.jl:3617-3626
    limw = [0.0] * S

    def limit_site(j, vnew, fn):                                                   # vasim.jl:1258-1330
        lb = mod.limit_sites[j]
        w = val(fn(val(vnew), vold[lb]))
        limw[j] = w
        ctx.record_limit_w(lidx[lb], w)
        seed = np.zeros(W)
        seed[N + j] = 1.0
        return vnew - val(vnew) + w + Dual(0.0, seed)

    def lim_delta(j):                                                              # limit_rhs_terms vasim.jl:2957-2966
        pl, nl = mod.limit_branches[mod.limit_sites[j]]
        return (Vf[pl] if pl >= 0 else 0.0) - (Vf[nl] if nl >= 0 else 0.0) - limw[j]

    temp_k = float(getattr(spec, "temp", 27.0)) + 273.15
    Ibr = evaluate(mod, Vd, par, temp_k, mfactor, spec.gmin if gmin is None else gmin, limit_site, ctx.initjct)
    for b, (pl, nl) in enumerate(mod.branches):
        p_node = node[pl] if pl >= 0 else 0
        n_node = node[nl] if nl >= 0 else 0
        I_branch = mfactor * Ibr[b]
        if isinstance(I_branch, CDual):
            I_resist, I_react, has_reactive = I_branch.r, I_branch.q, True
        else:
            I_resist, I_react, has_reactive = I_branch, 0.0, False
        # "determined by TYPE, not value" (vasim.jl:3388-3391): a branch whose contributions carry ddt() is reactive even
        # when this evaluation took a path without it
        has_reactive = has_reactive or mod.reactive[b]
        I_val, dI = val(I_resist), partials(I_resist, W)
        q_val, dq = val(I_react), partials(I_react, W)
        for k in range(N):
            k_node = node[k]
            if p_node != 0 and k_node != 0:
                ctx.stamp_G(p_node, k_node, dI[k])
            if n_node != 0 and k_node != 0:
                ctx.stamp_G(n_node, k_node, -dI[k])
        if has_reactive:
            V_branch = (Vf[pl] if pl >= 0 else 0.0) - (Vf[nl] if nl >= 0 else 0.0)
            name = "%s_%s_Q_%s_%s" % (instance, mod.name, mod.nodes[pl] if pl >= 0 else "0", mod.nodes[nl] if nl >= 0 else "0")
            if ctx.detect_or_cached(name, V_branch, q_val):
                qi = ctx.alloc_charge(name, p_node, n_node)
                if p_node != 0:
                    ctx.stamp_C(p_node, qi, 1.0 / CHARGE_SCALE)
                if n_node != 0:
                    ctx.stamp_C(n_node, qi, -1.0 / CHARGE_SCALE)
                ctx.stamp_G(qi, qi, 1.0)
                for k in range(N):
                    if node[k] != 0:
                        ctx.stamp_G(qi, node[k], -CHARGE_SCALE * dq[k])
                b_con = q_val
                for k in range(N):
                    b_con -= dq[k] * Vf[k]
                for j in range(S):
                    b_con += dq[N + j] * lim_delta(j)
                ctx.stamp_b(qi, CHARGE_SCALE * b_con)
            else:
                for k in range(N):
                    k_node = node[k]
                    if p_node != 0 and k_node != 0:
                        ctx.stamp_C(p_node, k_node, dq[k])
                    if n_node != 0 and k_node != 0:
                        ctx.stamp_C(n_node, k_node, -dq[k])
        Ieq = I_val
        for k in range(N):
            Ieq = Ieq + (-dI[k] * Vf[k])
        for j in range(S):
            Ieq = Ieq + dI[N + j] * lim_delta(j)
        if p_node != 0:
            ctx.stamp_b(p_node, -Ieq)
        if n_node != 0:
            ctx.stamp_b(n_node, Ieq)
