"""Host-side value evaluation of a parsed Verilog-A module (no derivatives).

Structure discovery needs the branch *charges* at a few probe points to decide, per reactive branch, between the
charge-state formulation and constant-capacitance stamps -- the reference does this by running its builder five times
and comparing apparent capacitances Q/V (build_with_detection, /root/reference/src/mna/solve.jl:1793-1822;
detect_or_cached!, /root/reference/src/mna/contrib.jl:214-257).  Parameter defaults are evaluated here as well.
"""
import math

from .frontend import VAError

K_BOLTZ, Q_ELEM = 1.380649e-23, 1.602176634e-19      # $vt = k T / q


class _Pair:
    """(resistive, reactive) value of an expression that contains ddt(): ddt(x) = (0, x)  (contrib.jl:356-375)."""
    __slots__ = ("r", "q")

    def __init__(self, r, q):
        self.r, self.q = r, q


def _limexp(x):
    return math.exp(x) if x < 80.0 else math.exp(80.0) * (1.0 + x - 80.0)


_F = {"exp": math.exp, "ln": math.log, "log": math.log10, "sqrt": math.sqrt, "pow": math.pow, "abs": abs, "min": min, "max": max,
      "limexp": _limexp, "tanh": math.tanh, "sinh": math.sinh, "cosh": math.cosh, "sin": math.sin, "cos": math.cos, "atan": math.atan}


def _r(x):
    return x.r if isinstance(x, _Pair) else x


def _q(x):
    return x.q if isinstance(x, _Pair) else 0.0


def evaluate(m, V, par, temp_k=300.15, mfactor=1.0, gmin=1e-12, vold=None, initjct=0):
    """Branch values of module ``m`` at node voltages ``V`` (list over m.nodes): ``[(I_b, q_b)]`` per branch, before the
    multiplicity factor.  ``par``: parameter name -> number (all of them; see ``defaults``); ``vold``: the value of the
    limit unknown of every $limit probe branch (zeros when omitted)."""
    env = {v: 0.0 for v in m.locals_}
    acc = [_Pair(0.0, 0.0) for _ in m.branches]
    vold = list(vold) if vold is not None else [0.0] * len(m.limit_branches)
    scope = [None]          # the analog function being evaluated: its variables shadow everything

    def call(fname, args):
        f_in, f_loc, f_body = m.functions[fname]
        saved = scope[0]
        scope[0] = dict({v: 0.0 for v in f_loc}, **dict(zip(f_in, args)), **{fname: 0.0})
        fenv = scope[0]
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
            a, b = m.node_index(e[1]), m.node_index(e[2])
            vnew = (V[a] if a >= 0 else 0.0) - (V[b] if b >= 0 else 0.0)
            return call(e[3], [vnew, vold[m.limit_sites[e[5][0]]]] + [ev(x) for x in e[4]])
        if k == "V":
            a, b = m.node_index(e[1]), m.node_index(e[2])
            return (V[a] if a >= 0 else 0.0) - (V[b] if b >= 0 else 0.0)
        if k == "ddt":
            return _Pair(0.0, _r(ev(e[1])))
        if k == "un":
            x = ev(e[2])
            if e[1] == "!":
                return 0.0 if x else 1.0
            return _Pair(-x.r, -x.q) if isinstance(x, _Pair) else -x
        if k == "cond":
            return ev(e[2]) if ev(e[1]) else ev(e[3])
        if k == "call":
            return _F[e[1]](*[ev(a) for a in e[2]])
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
                return float(initjct)
            if len(e[2]) > 1:
                return ev(e[2][1])
            raise VAError("$simparam(%r) has no value here" % (e[2][0][1] if e[2] else ""))
        op, l, r = e[1], ev(e[2]), ev(e[3])
        if op in ("+", "-"):
            s = 1.0 if op == "+" else -1.0
            if isinstance(l, _Pair) or isinstance(r, _Pair):
                return _Pair(_r(l) + s * _r(r), _q(l) + s * _q(r))
            return l + s * r
        if op == "*":
            if isinstance(l, _Pair):
                return _Pair(l.r * r, l.q * r)
            if isinstance(r, _Pair):
                return _Pair(l * r.r, l * r.q)
            return l * r
        if op == "/":
            return _Pair(l.r / r, l.q / r) if isinstance(l, _Pair) else l / r
        return float({"==": l == r, "!=": l != r, "<": l < r, ">": l > r, "<=": l <= r, ">=": l >= r,
                      "&&": bool(l) and bool(r), "||": bool(l) or bool(r)}[op])

    def run(stmts):
        for s in stmts:
            if s[0] == "assign":
                (scope[0] if scope[0] is not None else env)[s[1]] = ev(s[2])
            elif s[0] == "contrib":
                b = m.branches.index((m.node_index(s[1]), m.node_index(s[2])))
                x = ev(s[3])
                acc[b] = _Pair(acc[b].r + _r(x), acc[b].q + _q(x))
            elif s[0] == "block":
                run(s[1])
            elif s[0] == "if":
                run([s[2]] if ev(s[1]) else [s[3]])

    run(m.body)
    return [(a.r, a.q) for a in acc]


def static_eval(e, par, temp_k=300.15, mfactor=1.0, gmin=1e-12):
    """Value of an expression over parameters and numbers only (conditions that guard a node collapse)."""
    k = e[0]
    if k == "num":
        return e[1]
    if k == "var":
        return par[e[1]]
    if k == "un":
        return -static_eval(e[2], par) if e[1] == "-" else float(not static_eval(e[2], par))
    if k == "cond":
        return static_eval(e[2], par) if static_eval(e[1], par) else static_eval(e[3], par)
    if k == "call":
        return _F[e[1]](*[static_eval(a, par) for a in e[2]])
    if k == "sys":
        if e[1] == "$temperature":
            return temp_k
        if e[1] == "$vt":
            return K_BOLTZ * (static_eval(e[2][0], par) if e[2] else temp_k) / Q_ELEM
        if e[1] == "$mfactor":
            return mfactor
        if e[2] and e[2][0] == ("str", "gmin"):
            return gmin
        return static_eval(e[2][1], par)
    op, l, r = e[1], static_eval(e[2], par), static_eval(e[3], par)
    if op in "+-*/":
        return l + r if op == "+" else l - r if op == "-" else l * r if op == "*" else l / r
    return float({"==": l == r, "!=": l != r, "<": l < r, ">": l > r, "<=": l <= r, ">=": l >= r,
                  "&&": bool(l) and bool(r), "||": bool(l) or bool(r)}[op])


def defaults(m, given=None):
    """All parameter values of an instance: ``given`` overrides, the rest from the declarations (which may refer to
    earlier parameters)."""
    given = {k.lower(): v for k, v in (given or {}).items()}
    unknown = set(given) - {p.lower() for p in m.params}
    if unknown:
        raise VAError("%s has no parameter %s" % (m.name, ", ".join(sorted(unknown))))
    par = {}
    for name, expr in m.params.items():
        if name.lower() in given:
            par[name] = given[name.lower()]
            continue

        def ev(e):
            k = e[0]
            if k == "num":
                return e[1]
            if k == "var":
                return par[e[1]]
            if k == "un":
                return -ev(e[2]) if e[1] == "-" else float(not ev(e[2]))
            if k == "call":
                return _F[e[1]](*[ev(a) for a in e[2]])
            if k == "bin" and e[1] in "+-*/":
                l, r = ev(e[2]), ev(e[3])
                return l + r if e[1] == "+" else l - r if e[1] == "-" else l * r if e[1] == "*" else l / r
            raise VAError("%s: unsupported construct in the default of %s" % (m.name, name))
        par[name] = ev(expr)
    return par
