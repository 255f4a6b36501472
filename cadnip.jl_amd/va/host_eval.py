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


def _trunc(x):
    return float(math.trunc(x))


_F = {"exp": math.exp, "ln": math.log, "log": math.log10, "sqrt": math.sqrt, "pow": math.pow, "abs": abs, "min": min, "max": max,
      "limexp": _limexp, "tanh": math.tanh, "sinh": math.sinh, "cosh": math.cosh, "sin": math.sin, "cos": math.cos, "atan": math.atan,
      "tan": math.tan, "asin": math.asin, "acos": math.acos, "atan2": math.atan2, "hypot": math.hypot, "asinh": math.asinh,
      "acosh": math.acosh, "atanh": math.atanh, "floor": lambda x: float(math.floor(x)), "ceil": lambda x: float(math.ceil(x)), "int": _trunc}

# $simparam(name[, default]): MNASpec fields resolve (vasim.jl:1190-1218), the rest take their default
SPEC_SIMPARAMS = {"tnom": 27.0, "gshunt": 0.0, "srcFact": 1.0, "abstol": 1e-12, "reltol": 1e-3, "vntol": 1e-6, "iabstol": 1e-12, "time": 0.0}


def _r(x):
    return x.r if isinstance(x, _Pair) else x


def _q(x):
    return x.q if isinstance(x, _Pair) else 0.0


def evaluate(m, V, par, temp_k=300.15, mfactor=1.0, gmin=1e-12, vold=None, initjct=0, given=None, mode="tran", simparams=None,
             on_short=None, on_contrib=None, on_noise=None):
    """Branch values of module ``m`` at node voltages ``V`` (list over m.nodes): ``[(I_b, q_b)]`` per branch, before the
    multiplicity factor.  ``par``: parameter name -> number (all of them; see ``defaults``); ``vold``: the value of the
    limit unknown of every $limit probe branch (zeros when omitted); ``given``: the parameters the instance sets explicitly
    ($param_given; default: all of ``par``); ``on_short(a, b, stmt)``: called for every V(a,b) <+ 0 that executes; ``on_contrib(branch)``: for every
    executed current contribution; ``on_noise(a, b, fn, pwr, expo, label)``: for every white_noise / flicker_noise call inside the contribution to
    (a, b) -- the source it registers between those nodes, power already scaled by the multiplicity factor (vasim.jl:2856-2893)."""
    given = set(par) if given is None else {g for g in given}
    for al, target in m.aliasparams.items():
        if al in given:
            given.add(target)
    sp = dict(SPEC_SIMPARAMS, temp=temp_k - 273.15)
    if simparams:
        sp.update(simparams)
    env = {v: 0.0 for v in m.locals_}
    acc = [_Pair(0.0, 0.0) for _ in m.branches]
    bound = [None]       # (p, n) of the contribution being evaluated: where a noise call in its right-hand side injects
    vold = list(vold) if vold is not None else [0.0] * len(m.limit_branches)
    scopes = []             # analog-function frames: the innermost shadows everything

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
        f_args, f_loc, f_body = m.functions[fname]
        dirs = m.func_dirs.get(fname) or ["in"] * len(f_args)
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

    def ev(e):
        k = e[0]
        if k in ("num", "str"):
            return e[1]
        if k == "var":
            return lookup(e[1])
        if k == "given":
            return 1.0 if e[1] in given else 0.0
        if k == "analysis":
            return float(any((a in ("dc", "static") and mode == "dcop") or (a in ("tran", "transient") and mode == "tran")
                             or (a == "ac" and mode == "ac") for a in e[1]))
        if k == "noise":
            if on_noise is not None and bound[0] is not None and e[1] in ("white_noise", "flicker_noise"):
                pwr = _r(ev(e[2][0])) if e[2] else 0.0
                expo = _r(ev(e[2][1])) if (e[1] == "flicker_noise" and len(e[2]) >= 2) else 1.0
                on_noise(bound[0][0], bound[0][1], e[1], mfactor * pwr, expo, e[3])
            return 0.0           # no current on this path
        if k in ("Iprobe", "ddx"):
            return 0.0           # ddx: a derivative read-out, not a value the stamps use
        if k == "ucall":
            return call(e[1], e[2])
        if k == "limit":
            a, b = m.node_index(e[1]), m.node_index(e[2])
            vnew = (V[a] if a >= 0 else 0.0) - (V[b] if b >= 0 else 0.0)
            return call(e[3], [("num", vnew), ("num", vold[m.limit_sites[e[5][0]]])] + [("num", _r(ev(x))) for x in e[4]])
        if k == "V":
            a, b = m.node_index(e[1]), m.node_index(e[2])
            return (V[a] if a >= 0 else 0.0) - (V[b] if b >= 0 else 0.0)
        if k == "ddt":
            return _Pair(0.0, _r(ev(e[1])))
        if k == "un":
            x = ev(e[2])
            if e[1] == "!":
                return 0.0 if _r(x) else 1.0
            if e[1] == "~":
                return float(~int(_r(x)))
            return _Pair(-x.r, -x.q) if isinstance(x, _Pair) else -x
        if k == "cond":
            return ev(e[2]) if _r(ev(e[1])) else ev(e[3])
        if k == "call":
            return _F[e[1]](*[_r(ev(a)) for a in e[2]])
        if k == "sys":
            if e[1] == "$table_model":
                return table_value(m, e, [_r(ev(a)) for a in e[2][:-2]])
            if e[1] == "$temperature":
                return temp_k
            if e[1] == "$vt":
                return K_BOLTZ * (_r(ev(e[2][0])) if e[2] else temp_k) / Q_ELEM
            if e[1] == "$mfactor":
                return mfactor
            if e[1] in ("$abstime", "$realtime"):
                return float(sp.get("time", 0.0))
            name = e[2][0][1] if e[2] and e[2][0][0] == "str" else None
            if name == "gmin":
                return gmin
            if name in ("initjct", "iniLim"):
                return float(initjct)
            if name in sp:
                return float(sp[name])
            if len(e[2]) > 1:
                return ev(e[2][1])
            raise VAError("$simparam(%r) has no value here" % name)
        op, l, r = e[1], ev(e[2]), ev(e[3])
        if op in ("+", "-"):
            s = 1.0 if op == "+" else -1.0
            if isinstance(l, _Pair) or isinstance(r, _Pair):
                return _Pair(_r(l) + s * _r(r), _q(l) + s * _q(r))
            return l + s * r
        if op == "*":
            if isinstance(l, _Pair) and isinstance(r, _Pair):
                return l.r * r.r
            if isinstance(l, _Pair):
                return _Pair(l.r * r, l.q * r)
            if isinstance(r, _Pair):
                return _Pair(l * r.r, l * r.q)
            return l * r
        if op == "/":
            return _Pair(l.r / _r(r), l.q / _r(r)) if isinstance(l, _Pair) else l / _r(r)
        l, r = _r(l), _r(r)
        if op == "%":
            return math.fmod(l, r)
        if op in ("&", "|", "^", "<<", ">>"):
            a, b = int(l), int(r)
            return float({"&": a & b, "|": a | b, "^": a ^ b, "<<": a << b, ">>": a >> b}[op])
        return float({"==": l == r, "!=": l != r, "<": l < r, ">": l > r, "<=": l <= r, ">=": l >= r,
                      "&&": bool(l) and bool(r), "||": bool(l) or bool(r)}[op])

    def run(stmts):
        for s in stmts:
            k = s[0]
            if k == "assign":
                store(s[1], ev(s[2]))
            elif k == "contrib":
                if on_contrib is not None and s[3][0] != "noise":
                    on_contrib(m.branches.index((m.node_index(s[1]), m.node_index(s[2]))))
                bound[0] = (m.node_index(s[1]), m.node_index(s[2]))
                if s[3][0] == "noise":
                    ev(s[3])
                    bound[0] = None
                    continue
                b = m.branches.index((m.node_index(s[1]), m.node_index(s[2])))
                x = ev(s[3])
                bound[0] = None
                acc[b] = _Pair(acc[b].r + _r(x), acc[b].q + _q(x))
            elif k == "block":
                run(s[1])
            elif k == "if":
                run([s[2]] if _r(ev(s[1])) else [s[3]])
            elif k == "case":
                sel = _r(ev(s[1]))
                chosen = None
                for vals, body in s[2]:
                    if vals is not None and any(_r(ev(v)) == sel for v in vals):
                        chosen = body
                        break
                if chosen is None:
                    chosen = next((body for vals, body in s[2] if vals is None), None)
                if chosen is not None:
                    run([chosen])
            elif k == "while":
                n = 0
                while _r(ev(s[1])):
                    run([s[2]])
                    n += 1
                    if n > 100000:
                        raise VAError("%s: while loop does not terminate" % m.name)
            elif k == "for":
                run([s[1]])
                n = 0
                while _r(ev(s[2])):
                    run([s[4]]); run([s[3]])
                    n += 1
                    if n > 100000:
                        raise VAError("%s: for loop does not terminate" % m.name)
            elif k == "callstmt":
                call(s[1], s[2])
            elif k == "fatal":
                raise VAError("%s: %s %s" % (m.name, s[1], s[2]))
            elif k == "short":
                if on_short is not None:
                    on_short(m.node_index(s[1]), m.node_index(s[2]), s)

    for name, ie in m.local_init:                               # module-scope initialisers, in declaration order
        env[name] = ev(ie)
    run(m.body)
    return [(a.r, a.q) for a in acc]


def instance_structure(m, par, given=None, temp_k=300.15, mfactor=1.0, gmin=1e-12):
    """What one evaluation at zero bias says about the structure of this instance (conditions decided by the parameters):

    * alias: internal node -> the node it is merged into (-1 = ground), for the executed V(a,b) <+ 0 statements of the alias
      kind (VAModule.short_is_alias: the reference's detect_short_circuits pattern, vasim.jl:2723-2818);
    * shorts_on[j]: the j-th statement of ``m.vshorts`` executes -- a potential contribution V(a,b) <+ 0 that owns a branch
      current (vasim.jl:2311-2395; skipped when both nets are one unknown already);
    * active[b]: branch b is stamped -- always, unless every contribution of it sits under parameter-decided conditions and
      none of them executes (the reference stamps such contributions inline, vasim.jl:2397-2470)."""
    np_ = len(m.ports)
    out = {}
    kind = {id(m.shorts[i][3]): (m.vshorts.index(i) if i in m.vshorts else -1) for i in range(len(m.shorts))}
    shorts_on = [False] * len(m.vshorts)
    touched = [False] * len(m.branches)

    def root(i):
        while i in out and i >= 0:
            i = out[i]
        return i

    def on_short(a, b, stmt):
        j = kind.get(id(stmt), -1)
        if j >= 0:
            shorts_on[j] = True
            return
        a, b = root(a), root(b)
        if a == b:
            return
        if a >= np_:
            out[a] = b
        elif b >= np_:
            out[b] = a
        else:
            raise VAError("%s: V(%s,%s) <+ 0 between two terminals needs a branch current (not supported)" % (m.name, m.nodes[a], m.nodes[b]))

    def on_contrib(b):
        touched[b] = True
    try:
        evaluate(m, [0.0] * m.n_nodes, par, temp_k, mfactor, gmin, given=given, on_short=on_short, on_contrib=on_contrib)
    except (ValueError, ZeroDivisionError, OverflowError):
        pass                    # a zero-bias probe may leave the model's domain after the collapse statements (setup section) ran
    alias = {k: root(k) for k in out}
    for j, si in enumerate(m.vshorts):
        if m.short_kind[si] != "cond":
            shorts_on[j] = True                  # top level of the analog block: executes for every instance
        if m.short_kind[si] == "named":
            continue                             # (a named branch is stamped unconditionally, vasim.jl:3700-3712)
        a, b = m.shorts[si][0], m.shorts[si][1]   # both nets already one unknown: nothing to stamp (vasim.jl:2364, 3765 `if p != n`)
        if shorts_on[j] and alias.get(a, a) == alias.get(b, b):
            shorts_on[j] = False
    active = [touched[b] or not m.branch_guarded[b] for b in range(len(m.branches))]
    return alias, shorts_on, active


def collapsed_nodes(m, par, given=None, temp_k=300.15, mfactor=1.0, gmin=1e-12):
    """internal node -> the node it is merged into (-1 = ground) for this instance (see instance_structure)."""
    return instance_structure(m, par, given, temp_k, mfactor, gmin)[0]


def table_value(m, e, xs):
    """value of the $table_model call ``e`` of module ``m`` at the inputs ``xs`` (the table file is looked for next to the module's source)"""
    import os
    from . import table_model
    fn = e[2][-2][1]
    path = fn if os.path.isabs(fn) or m.include_dir is None else os.path.join(m.include_dir, fn)
    return table_model.lookup(path, e[2][-1][1], xs)


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
    alias = {a.lower(): t.lower() for a, t in m.aliasparams.items()}
    given = {alias.get(k.lower(), k.lower()): v for k, v in (given or {}).items()}      # aliasparam a = p: a sets p
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
            if k == "str":
                return e[1]
            if k == "cond":
                return ev(e[2]) if ev(e[1]) else ev(e[3])
            if k == "bin":
                l, r = ev(e[2]), ev(e[3])
                return float({"==": l == r, "!=": l != r, "<": l < r, ">": l > r, "<=": l <= r, ">=": l >= r,
                              "&&": bool(l) and bool(r), "||": bool(l) or bool(r)}[e[1]])
            if k == "sys" and e[1] == "$simparam" and len(e[2]) > 1:
                return ev(e[2][1])
            raise VAError("%s: unsupported construct in the default of %s" % (m.name, name))
        par[name] = ev(expr)
    return par
