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
import os

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


def _res(x):
    """a value with a ddt() part enters nonlinear functions, comparisons and conditions with its resistive part: the primal
    slot of the reference's Dual{ContributionTag} (contrib.jl:356-375)"""
    return x.r if isinstance(x, CDual) else x


def _u(fn, dfn):
    g = _unary(fn, dfn)
    return lambda x: g(_res(x))


def _atan2(y, x):
    y, x = _res(y), _res(x)
    yv, xv = val(y), val(x)
    f = math.atan2(yv, xv)
    d = xv * xv + yv * yv
    p = 0.0
    if isinstance(y, Dual):
        p = p + y.p * (xv / d)
    if isinstance(x, Dual):
        p = p + x.p * (-yv / d)
    return Dual(f, p) if isinstance(y, Dual) or isinstance(x, Dual) else f


def _hypot(a, b):
    a, b = _res(a), _res(b)
    return dsqrt(a * a + b * b)


def _int_like(fn):
    return lambda x: float(fn(val(_res(x))))          # piecewise constant: no partials


FUNCS = {
    "exp": lambda x: dexp(_res(x)), "ln": lambda x: dln(_res(x)), "sqrt": lambda x: dsqrt(_res(x)), "abs": lambda x: dabs(_res(x)),
    "pow": lambda a, b: _pow(_res(a), _res(b)),
    "log": _u(math.log10, lambda v, f: 1.0 / (v * math.log(10.0))),
    "limexp": _u(_limexp, lambda v, f: f if v < 80.0 else math.exp(80.0)),
    "tanh": _u(math.tanh, lambda v, f: 1.0 - f * f), "sinh": _u(math.sinh, lambda v, f: math.cosh(v)),
    "cosh": _u(math.cosh, lambda v, f: math.sinh(v)), "sin": _u(math.sin, lambda v, f: math.cos(v)),
    "cos": _u(math.cos, lambda v, f: -math.sin(v)), "atan": _u(math.atan, lambda v, f: 1.0 / (1.0 + v * v)),
    "tan": _u(math.tan, lambda v, f: 1.0 + f * f), "asin": _u(math.asin, lambda v, f: 1.0 / math.sqrt(1.0 - v * v)),
    "acos": _u(math.acos, lambda v, f: -1.0 / math.sqrt(1.0 - v * v)), "asinh": _u(math.asinh, lambda v, f: 1.0 / math.sqrt(v * v + 1.0)),
    "acosh": _u(math.acosh, lambda v, f: 1.0 / math.sqrt(v * v - 1.0)), "atanh": _u(math.atanh, lambda v, f: 1.0 / (1.0 - v * v)),
    "atan2": _atan2, "hypot": _hypot,
    "floor": _int_like(math.floor), "ceil": _int_like(math.ceil), "int": _int_like(lambda v: math.trunc(v)),
    # ties take the first operand (the product's va_max / va_min)
    "max": lambda a, b: _res(b) if val(_res(b)) > val(_res(a)) else _res(a), "min": lambda a, b: _res(b) if val(_res(b)) < val(_res(a)) else _res(a),
}

# $simparam(name[, default]): names that are MNASpec fields resolve, everything else takes its default (vasim.jl:1190-1218)
SPEC_SIMPARAMS = ("temp", "gmin", "gshunt", "srcFact", "tnom", "abstol", "reltol", "vntol", "iabstol", "time")


class VAFatal(RuntimeError):
    pass


def table_model_value(path, ctrl, xs):
    """$table_model (src/vasim.jl:762-845 for the file and control string, src/mna/table_model.jl:49-58 for the interpolant: Interpolations.jl's
    gridded linear interpolation with Line / Flat / Throw extrapolation) at plain-number inputs."""
    interp, col = ctrl.split(";")
    dims = [d.strip() for d in interp.split(",")]
    assert len(dims) == len(xs) and all(d[:1] == "1" for d in dims), ctrl
    ex = {(d[1:] or "L") for d in dims}
    assert len(ex) == 1, ctrl
    ex = ex.pop()
    rows = [[float(t) for t in ln.split("#", 1)[0].split()] for ln in open(path) if ln.split("#", 1)[0].strip()]
    D = len(xs)
    axes = [sorted({r[k] for r in rows}) for k in range(D)]
    grid = {tuple(axes[k].index(r[k]) for k in range(D)): r[D + int(col) - 1] for r in rows}
    xc = [min(max(x, ax[0]), ax[-1]) for x, ax in zip(xs, axes)]
    if ex == "E" and xc != list(xs):
        raise ValueError("$table_model: outside the table")
    cell = []
    for x, ax in zip(xc, axes):
        i = max(k for k in range(len(ax) - 1) if ax[k] <= x)
        cell.append(i)

    def at(pt):
        fr = [(p - axes[d][cell[d]]) / (axes[d][cell[d] + 1] - axes[d][cell[d]]) for d, p in enumerate(pt)]
        v = 0.0
        for c in range(1 << D):
            w, idx = 1.0, []
            for d in range(D):
                hi = (c >> d) & 1
                w *= fr[d] if hi else 1.0 - fr[d]
                idx.append(cell[d] + hi)
            v += w * grid[tuple(idx)]
        return v
    v = at(xc)
    if ex == "L":
        for d in range(D):
            if xs[d] != xc[d]:
                lo, hi = list(xc), list(xc)
                lo[d], hi[d] = axes[d][cell[d]], axes[d][cell[d] + 1]
                v += (at(hi) - at(lo)) / (hi[d] - lo[d]) * (xs[d] - xc[d])
    return v


def evaluate(mod, Vd, par, temp_k, mfactor, gmin, limit_site=None, initjct=False, given=None, spec=None, on_short=None, touched=None, probe=None, on_noise=None):
    """Branch contributions of ``mod`` on dual node voltages ``Vd``: one Dual / CDual / float per branch.
    ``limit_site(j, vnew_dual, fn)`` implements a $limit call site (stamp_va); ``given``: the parameters the instance sets
    explicitly ($param_given); ``on_short(a, b, stmt, value)``: called for every executed potential contribution V(a,b) <+ value;
    ``touched``: a list that receives True at the index of every branch a current contribution executes for; ``probe(e)``: the value
    ``on_noise(a, b, fn, pwr, expo, label)``: a white_noise / flicker_noise call inside the contribution to (a, b) (vasim.jl:2856-2893:
    the call registers a source between the enclosing contribution's nodes, scaled by $mfactor; its value is 0.0); of a current probe:
    a plain number, the branch-current unknown of a potential contribution (vasim.jl:3632-3640,
    3652-3667) -- default 0.0 (branches that carry noise only, vasim.jl:3641-3650)."""
    given = set(par) if given is None else set(given)
    for al, target in mod.aliasparams.items():
        if al in given:
            given.add(target)
    env = {v: 0.0 for v in mod.locals_}
    acc = [0.0 for _ in mod.branches]
    scopes = []          # analog-function frames: the innermost shadows everything
    bound = [None]       # (p, n) of the contribution being evaluated
    mode = getattr(spec, "mode", "tran") if spec is not None else "tran"

    def lookup(name):
        if scopes:
            return scopes[-1][name]
        return par[name] if name in par else env[name]

    def store(name, value):
        if scopes:
            scopes[-1][name] = value
        else:
            env[name] = value

    def call(fname, arg_exprs):
        f_args, f_loc, f_body = mod.functions[fname]
        dirs = mod.func_dirs.get(fname) or ["in"] * len(f_args)
        vals = [ev(a) for a in arg_exprs]
        frame = dict({v: 0.0 for v in f_loc}, **dict(zip(f_args, vals)), **{fname: 0.0})
        scopes.append(frame)
        try:
            run(f_body)
        finally:
            scopes.pop()
        for a, d, nm in zip(arg_exprs, dirs, f_args):         # output / inout arguments are passed by reference
            if d != "in":
                store(a[1], frame[nm])
        return frame[fname]

    def simparam(e):
        name = e[2][0][1] if e[2] and e[2][0][0] == "str" else None
        if name in ("initjct", "iniLim"):                      # ngspice MODEINITJCT -> the PCNR initjct flag (vasim.jl:1198-1206)
            return 1.0 if initjct else 0.0
        if name == "gmin":
            return gmin
        if name in SPEC_SIMPARAMS and spec is not None and hasattr(spec, name):
            return float(getattr(spec, name))
        if len(e[2]) > 1:
            return ev(e[2][1])
        raise KeyError("Unknown simparam: %s" % name)

    def ev(e):
        k = e[0]
        if k == "num":
            return e[1]
        if k == "str":
            return e[1]
        if k == "var":
            return lookup(e[1])
        if k == "given":
            return 1.0 if (e[1] in given) else 0.0
        if k == "analysis":
            # vasim.jl:1220-1250: dc / static -> :dcop, tran / transient -> :tran, ac -> :ac, nodeset -> false
            return float(any((a in ("dc", "static") and mode == "dcop") or (a in ("tran", "transient") and mode == "tran")
                             or (a == "ac" and mode == "ac") for a in e[1]))
        if k == "noise":
            if on_noise is not None and bound[0] is not None and e[1] in ("white_noise", "flicker_noise"):
                pwr = val(_res(ev(e[2][0]))) if e[2] else 0.0
                expo = val(_res(ev(e[2][1]))) if (e[1] == "flicker_noise" and len(e[2]) >= 2) else 1.0
                on_noise(bound[0][0], bound[0][1], e[1], mfactor * pwr, expo, e[3])
            return 0.0                                          # noise sources contribute no current on this path
        if k == "Iprobe":
            return float(probe(e)) if probe is not None else 0.0
        if k == "ddx":
            x = _res(ev(e[1]))
            a = mod.node_index(e[2])
            d1 = float(x.p[a]) if isinstance(x, Dual) and a >= 0 else 0.0
            if e[3] is None:
                return d1
            b = mod.node_index(e[3])                 # d/dV(a,b) = (d/dV_a - d/dV_b) / 2   (vasim.jl:1168-1180)
            d2 = float(x.p[b]) if isinstance(x, Dual) and b >= 0 else 0.0
            return (d1 - d2) / 2
        if k == "ucall":
            return call(e[1], e[2])
        if k == "limit":
            a, b = mod.node_index(e[1]), mod.node_index(e[2])
            vnew = (Vd[a] if a >= 0 else 0.0) - (Vd[b] if b >= 0 else 0.0)
            user = [_res(ev(x)) for x in e[4]]
            fn = e[3]
            return limit_site(e[5][0], vnew, lambda vn, vo: call(fn, [("num", vn), ("num", vo)] + [("num", val(u)) for u in user]))
        if k == "V":
            a, b = mod.node_index(e[1]), mod.node_index(e[2])
            return (Vd[a] if a >= 0 else 0.0) - (Vd[b] if b >= 0 else 0.0)
        if k == "ddt":
            return va_ddt(ev(e[1]))
        if k == "un":
            x = ev(e[2])
            if e[1] == "!":
                return 0.0 if val(_res(x)) else 1.0
            if e[1] == "~":
                return float(~int(val(_res(x))))
            return -x
        if k == "cond":
            return ev(e[2]) if val(_res(ev(e[1]))) else ev(e[3])
        if k == "call":
            return FUNCS[e[1]](*[ev(a) for a in e[2]])
        if k == "sys":
            if e[1] == "$table_model":
                fn = e[2][-2][1]
                path = fn if (os.path.isabs(fn) or getattr(mod, "include_dir", None) is None) else os.path.join(mod.include_dir, fn)
                return table_model_value(path, e[2][-1][1], [val(_res(ev(a))) for a in e[2][:-2]])
            if e[1] == "$temperature":
                return temp_k
            if e[1] == "$vt":
                return K_BOLTZ * (val(_res(ev(e[2][0]))) if e[2] else temp_k) / Q_ELEM
            if e[1] == "$mfactor":
                return mfactor
            if e[1] in ("$abstime", "$realtime"):
                return float(getattr(spec, "time", 0.0)) if spec is not None else 0.0
            return simparam(e)
        op, l, r = e[1], ev(e[2]), ev(e[3])
        if isinstance(l, str) or isinstance(r, str):          # string parameters compare as strings (bsim4v8: version == "4.8.3")
            if op == "==":
                return float(str(l) == str(r))
            if op == "!=":
                return float(str(l) != str(r))
            raise ValueError("%s: operator %s on a string" % (mod.name, op))
        if op == "+":
            return l + r
        if op == "-":
            return l - r
        if op == "*":
            if isinstance(l, CDual) and isinstance(r, CDual):
                return l.r * r.r                                # (no model multiplies two ddt() terms: resistive parts)
            return l * r
        if op == "/":
            return l / _res(r)
        lv, rv = val(_res(l)), val(_res(r))
        if op == "%":
            return math.fmod(lv, rv)
        if op in ("&", "|", "^", "<<", ">>"):
            a, b = int(lv), int(rv)
            return float({"&": a & b, "|": a | b, "^": a ^ b, "<<": a << b, ">>": a >> b}[op])
        return float({"==": lv == rv, "!=": lv != rv, "<": lv < rv, ">": lv > rv, "<=": lv <= rv, ">=": lv >= rv,
                      "&&": bool(lv) and bool(rv), "||": bool(lv) or bool(rv)}[op])

    def run(stmts):
        for s in stmts:
            k = s[0]
            if k == "assign":
                store(s[1], ev(s[2]))
            elif k == "contrib":
                bound[0] = (mod.node_index(s[1]), mod.node_index(s[2]))     # the branch a noise call in this right-hand side injects into
                if s[3][0] == "noise":
                    ev(s[3])
                    bound[0] = None
                    continue
                b = mod.branches.index((mod.node_index(s[1]), mod.node_index(s[2])))
                if touched is not None:
                    touched[b] = True
                acc[b] = acc[b] + ev(s[3])
                bound[0] = None
            elif k == "block":
                run(s[1])
            elif k == "if":
                run([s[2]] if val(_res(ev(s[1]))) else [s[3]])
            elif k == "case":
                sel = val(_res(ev(s[1])))
                chosen = None
                for vals, body in s[2]:
                    if vals is not None and any(val(_res(ev(v))) == sel for v in vals):
                        chosen = body
                        break
                if chosen is None:
                    chosen = next((body for vals, body in s[2] if vals is None), None)
                if chosen is not None:
                    run([chosen])
            elif k == "while":
                n = 0
                while val(_res(ev(s[1]))):
                    run([s[2]])
                    n += 1
                    if n > 100000:
                        raise RuntimeError("while loop does not terminate")
            elif k == "for":
                run([s[1]])
                n = 0
                while val(_res(ev(s[2]))):
                    run([s[4]]); run([s[3]])
                    n += 1
                    if n > 100000:
                        raise RuntimeError("for loop does not terminate")
            elif k == "callstmt":
                call(s[1], s[2])
            elif k == "fatal":
                raise VAFatal("%s: %s %s" % (mod.name, s[1], s[2]))
            elif k == "short":
                if on_short is not None:
                    on_short(mod.node_index(s[1]), mod.node_index(s[2]), s, ev(s[3]))

    for name, ie in mod.local_init:                             # module-scope initialisers, in declaration order
        env[name] = ev(ie)
    run(mod.body)
    return acc


def short_aliases_a_terminal(mod, stmt):
    """detect_short_circuits (vasim.jl:2723-2818): only ``V(int, ext) <+ 0`` in the if-branch of a top-level conditional aliases
    the internal node to the terminal.  Every other two-net V(a,b) <+ 0 is a potential contribution with a branch current
    (vasim.jl:2311-2395).  (The one-net form V(a) <+ 0 stays this build's alias to ground.)"""
    a, b, guards = next(x[:3] for x in mod.shorts if x[3] is stmt)
    np_ = len(mod.ports)
    if stmt[3] != ("num", 0.0) or stmt[4] is not None:
        return False              # a value, or a named branch: always a branch with its own current (vasim.jl:3229-3246)
    if b < 0:
        return a >= np_
    return len(guards) == 1 and guards[0][1] is True and ((a >= np_) != (b >= np_))


def collapsed_nodes(mod, par, given, spec, mfactor=1.0, gmin=1e-12):
    """Which internal nodes does this instance collapse?  The V(a,b) <+ 0 statements sit under conditions decided by the
    parameters (vasim.jl:2313-2395, 3533-3564); one evaluation of the analog block on plain numbers at zero bias executes
    exactly the ones that apply.  Returns internal node -> node it is merged into (-1 = ground)."""
    np_ = len(mod.ports)
    out = {}

    def root(i):
        while i in out and i >= 0:
            i = out[i]
        return i

    def on_short(a, b, stmt, value=0.0):
        if not short_aliases_a_terminal(mod, stmt):
            return
        a, b = root(a), root(b)
        if a == b:
            return
        if a >= np_:
            out[a] = b
        elif b >= np_:
            out[b] = a
        else:
            raise ValueError("%s: V(%s,%s) <+ 0 between two terminals" % (mod.name, mod.nodes[a], mod.nodes[b]))
    temp_k = float(getattr(spec, "temp", 27.0)) + 273.15
    try:
        evaluate(mod, [0.0] * len(mod.nodes), par, temp_k, mfactor, gmin, lambda j, vnew, fn: vnew, False, given, spec, on_short)
    except (ValueError, ZeroDivisionError, OverflowError, VAFatal):
        pass                      # a zero-bias probe may leave a model's domain after the collapse statements (the setup section) ran
    return {k: root(k) for k in out}


def stamp_va(ctx, mod, ext_nodes, x, par, spec, instance, mfactor=1.0, gmin=None, given=None):
    """The generated stamp! body for one instance of ``mod`` (vasim.jl:3886-3963)."""
    N, S = len(mod.nodes), mod.n_sites
    W = N + S
    # internal node allocation with short-circuit aliasing (vasim.jl:3533-3564): V(a,b) <+ 0 under a parameter condition
    alias = collapsed_nodes(mod, par, given, spec, mfactor, spec.gmin if gmin is None else gmin)
    node = list(ext_nodes) + [None] * (N - len(mod.ports))
    for k in range(len(mod.ports), N):
        if k not in alias:
            node[k] = ctx.alloc_internal_node("%s_%s_%s" % (instance, mod.name, mod.nodes[k]))
    for k in range(len(mod.ports), N):
        if k in alias:
            node[k] = node[alias[k]] if alias[k] >= 0 else 0
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

    # Potential contributions at the top level of the analog block own their branch currents from the start of the call: named
    # branches first, then two-node ones (branch_current_alloc, vasim.jl:3253-3280); their stamps follow the branches (below).
    def where(stmt):
        return next(i for i, x in enumerate(mod.shorts) if x[3] is stmt)
    top_cur, top_val = {}, {}
    for kind in ("named", "top"):
        for i in mod.vshorts:
            if mod.short_kind[i] == kind:
                a, b, st = mod.shorts[i][0], mod.shorts[i][1], mod.shorts[i][3]
                pn = "%s_%s" % (mod.nodes[a], mod.nodes[b] if b >= 0 else "0")
                top_cur[i] = ctx.alloc_current("%s_%s_I_%s" % (instance, mod.name, st[4]) if kind == "named" else "%s_%s_I_V_%s" % (instance, mod.name, pn))

    def probe(e):
        j = mod.probe_short(e)
        return 0.0 if j is None else x_at(x, ctx.resolve_index(top_cur[mod.vshorts[j]]))

    def twonode_stamps(iv, a, b, value):
        """vasim.jl:2363-2393 / 3765-3812: KCL columns, the constraint row with -dX/dV_k in every node column, b = X - sum dX/dV_k V_k"""
        p_node, n_node = (node[a] if a >= 0 else 0), (node[b] if b >= 0 else 0)
        v_val, dv = val(_res(value)), partials(_res(value), W)
        ctx.stamp_G(p_node, iv, 1.0)
        ctx.stamp_G(n_node, iv, -1.0)
        ctx.stamp_G(iv, p_node, 1.0)
        ctx.stamp_G(iv, n_node, -1.0)
        for k in range(N):
            ctx.stamp_G(iv, node[k], -dv[k])
        b_v = v_val
        for k in range(N):
            b_v -= dv[k] * Vf[k]
        ctx.stamp_b(iv, b_v)

    def on_short(a, b, stmt, value=0.0):
        """An executed potential contribution that is not a terminal alias.  Inside a conditional it is stamped where it stands, with
        its own branch current (vasim.jl:2363-2393); at the top level its value is kept for the stamps behind the branches."""
        if short_aliases_a_terminal(mod, stmt):
            return
        i = where(stmt)
        if mod.short_kind[i] != "cond":
            top_val[i] = value
            return
        p_node, n_node = (node[a] if a >= 0 else 0), (node[b] if b >= 0 else 0)
        if p_node == n_node:
            return
        iv = ctx.alloc_current("%s_I_V_%s_%s" % (instance, mod.nodes[a], mod.nodes[b] if b >= 0 else "0"))
        twonode_stamps(iv, a, b, value)

    temp_k = float(getattr(spec, "temp", 27.0)) + 273.15
    touched = [False] * len(mod.branches)
    def on_noise(a, b, fn, pwr, expo, label):             # noise_source_name(instance, label), context.jl:1123-1127
        p_node, n_node = (node[a] if a >= 0 else 0), (node[b] if b >= 0 else 0)
        name = ("%s_%s" % (instance, label)) if (instance and label) else (instance or label or "va")
        if fn == "white_noise":
            ctx.register_white_noise(p_node, n_node, pwr, name)
        else:
            ctx.register_flicker_noise(p_node, n_node, pwr, expo, name)

    Ibr = evaluate(mod, Vd, par, temp_k, mfactor, spec.gmin if gmin is None else gmin, limit_site, ctx.initjct, given, spec, on_short, touched, probe,
                   on_noise if hasattr(ctx, "register_white_noise") else None)
    for b, (pl, nl) in enumerate(mod.branches):
        if mod.branch_guarded[b] and not touched[b]:
            continue          # contributions inside conditionals are stamped inline, when they execute (vasim.jl:2397-2470): none did
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
    # named branches V(br) <+ X (vasim.jl:3669-3746): no partials -- b[I] = the resistive value, C[I,I] = -(the value under ddt())
    for i in mod.vshorts:
        if mod.short_kind[i] == "named":
            a, b, iv = mod.shorts[i][0], mod.shorts[i][1], top_cur[i]
            p_node, n_node = (node[a] if a >= 0 else 0), (node[b] if b >= 0 else 0)
            X = top_val.get(i, 0.0)
            ctx.stamp_G(p_node, iv, 1.0)
            ctx.stamp_G(n_node, iv, -1.0)
            ctx.stamp_G(iv, p_node, 1.0)
            ctx.stamp_G(iv, n_node, -1.0)
            if isinstance(X, CDual):
                ctx.stamp_b(iv, val(X.r))
                ctx.stamp_C(iv, iv, -val(X.q))
            else:
                ctx.stamp_b(iv, val(X))
    # two-node potential contributions at the top level (vasim.jl:3750-3815)
    for i in mod.vshorts:
        if mod.short_kind[i] == "top":
            a, b = mod.shorts[i][0], mod.shorts[i][1]
            if (node[a] if a >= 0 else 0) != (node[b] if b >= 0 else 0):
                twonode_stamps(top_cur[i], a, b, top_val.get(i, 0.0))
