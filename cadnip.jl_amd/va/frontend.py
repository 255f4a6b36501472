"""Verilog-A front end for compact models: text -> ``VAModule``.

The reference compiles a Verilog-A module into a ``stamp!`` method: every ``I(p,n) <+ expr`` becomes a branch current
evaluated on forward-mode duals, ``ddt(x)`` turns the contribution into a (resistive, reactive) pair
(/root/reference/src/vasim.jl:2993-3985, /root/reference/src/mna/contrib.jl:356-375).  This module parses the subset
those compact models are written in and does the static analysis the code generators need; ``hipgen.py`` emits the
MI355X stamp function from it, ``host_eval.py`` evaluates branch charges on the host (voltage-dependence detection).

Supported::

    module NAME (ports);  inout|input|output ...;  electrical a, b, c;      // electrical nets outside the port list are internal nodes
    parameter real|integer P = expr [from range];   real|integer x, y;
    analog begin ... end
      x = expr;   I(a,b) <+ expr;   I(a) <+ expr;   if (c) stmt [else stmt];   begin ... end
    expressions: + - * / unary- ! comparison && || ?:  numbers with scale factors (T G M K k m u n p f a)
      V(a,b) V(a)  ddt(e)  exp ln log sqrt pow abs min max limexp tanh sinh cosh sin cos atan
      $vt [$vt(T)]  $temperature  $mfactor  $simparam("gmin"|"initjct"[, default])
    analog function real NAME; input a, b; real x; begin ... NAME = expr; end endfunction      // pure functions of their inputs
    $limit(V(p,n), NAME, args...)      // PCNR limiting with the user function NAME(vnew, vold, args...) (vasim.jl:1258-1330);
                                       // top level of the analog block only, as in the reference

    if (static condition) V(a,b) <+ 0; else I(a,b) <+ ...      // node collapse: the internal node of the pair is aliased
                                       // to the other one for instances whose parameters make the condition true

    branch (a,b) name;   V(name)  I(name) <+ ...                 // named branches

Not supported (an error, never a silent approximation): other potential contributions ``V() <+ expr``,
``@(...)`` events, loops, ``idt``, noise sources, the string form of ``$limit``.
"""
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

_SCALE = {"T": 1e12, "G": 1e9, "M": 1e6, "K": 1e3, "k": 1e3, "m": 1e-3, "u": 1e-6, "n": 1e-9, "p": 1e-12, "f": 1e-15, "a": 1e-18}
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?[TGMKkmunpfa]?(?![A-Za-z_0-9]))
  | (?P<id>\$?[A-Za-z_][A-Za-z_0-9$]*)
  | (?P<str>"[^"]*")
  | (?P<op><\+|==|!=|<=|>=|&&|\|\||[-+*/()<>!?:;,=@\[\]{}])
""", re.X | re.S)

FUNCS = {"exp": 1, "ln": 1, "log": 1, "sqrt": 1, "pow": 2, "abs": 1, "min": 2, "max": 2, "limexp": 1, "tanh": 1, "sinh": 1,
         "cosh": 1, "sin": 1, "cos": 1, "atan": 1}


class VAError(ValueError):
    pass


def tokenize(text):
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("`"))   # `include / `define lines
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise VAError("cannot tokenize at %r" % text[pos:pos + 30])
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group(m.lastgroup)))
    out.append(("eof", ""))
    return out


@dataclass
class VAModule:
    name: str
    ports: List[str]
    nodes: List[str]                       # ports, then internal nodes (declaration order)
    params: Dict[str, tuple]               # name -> default expression (AST), in declaration order
    locals_: List[str]
    body: list                             # statements
    branches: List[Tuple[int, int]] = field(default_factory=list)     # (p, n) node indices, -1 = ground; first-use order
    reactive: List[bool] = field(default_factory=list)                # per branch: its contributions carry ddt()
    var_is_dual: Dict[str, bool] = field(default_factory=dict)        # local depends on a voltage
    var_is_reactive: Dict[str, bool] = field(default_factory=dict)    # local carries a ddt() part
    functions: Dict[str, tuple] = field(default_factory=dict)         # analog functions: name -> (inputs, locals, body)
    limit_branches: List[Tuple[int, int]] = field(default_factory=list)   # probe branches of the $limit sites, first-use order
    limit_sites: List[int] = field(default_factory=list)              # per $limit call site (evaluation order): its limit branch
    shorts: List[tuple] = field(default_factory=list)                 # V(a,b) <+ 0: (a, b, [(static condition, wanted truth)])
    source: str = ""

    @property
    def n_nodes(self):
        return len(self.nodes)

    @property
    def n_internal(self):
        return len(self.nodes) - len(self.ports)

    @property
    def n_sites(self):
        return len(self.limit_sites)

    def aliases(self, par):
        """Node collapse of one instance: internal node index -> the node it is merged into, for the ``V(a,b) <+ 0``
        statements whose (parameter-only) conditions hold with the parameter values ``par``."""
        from .host_eval import static_eval
        out = {}
        np_ = len(self.ports)

        def root(i):
            while i in out:
                i = out[i]
            return i
        for a, b, guards in self.shorts:
            if all(bool(static_eval(c, par)) == want for c, want in guards):
                a, b = root(a), root(b)
                if a == b:
                    continue
                if a >= np_:
                    out[a] = b
                elif b >= np_:
                    out[b] = a
                else:
                    raise VAError("%s: V(%s,%s) <+ 0 between two terminals needs a branch current (not supported)"
                                  % (self.name, self.nodes[a], self.nodes[b]))
        return {k: root(k) for k in out}

    # ---- stamp layout (structure.py / hipgen.py / oracle agree on it) ------------------------------------------
    # local unknowns: nodes 0..N-1, then one charge unknown per branch (ground when the branch has none), then one limit
    # unknown per $limit probe branch;  g_lim rows (l,l) (l,p) (l,n) of limit branch l -> G slots 2N B + (N+1) B + 3 l + {0,1,2}
    # G slots: branch b, node k:  (p,k) -> 2N b + 2k, (n,k) -> 2N b + 2k + 1;  charge row of branch b: (q,q) -> 2N B + (N+1) b,
    #          (q,k) -> 2N B + (N+1) b + 1 + k
    # C slots: charge columns (p,q) -> 2b, (n,q) -> 2b + 1;  linear form (p,k) -> 2B + 2N b + 2k, (n,k) -> ... + 1
    # b slots: branch b: p -> 3b, n -> 3b + 1, charge row -> 3b + 2
    def shape(self):
        N, B, L = self.n_nodes, len(self.branches), len(self.limit_branches)
        n_par = len(self.params) + 3        # + temperature [K], mfactor, gmin
        return (N + B + L, 2 * N * B + (N + 1) * B + 3 * L, 2 * B + 2 * N * B, 3 * B, n_par, 2)

    def program(self, vdep):
        """(stream, local slot, local row, local col) in the reference's stamp order (vasim.jl:3374-3521): per branch the
        resistive Jacobian, then the reactive part in charge-state or linear form, then the equivalent currents."""
        N, B = self.n_nodes, len(self.branches)
        prog = []
        for l, (p, n) in enumerate(self.limit_branches):     # the hoisted $limit preamble (vasim.jl:3110-3138)
            ul, g0 = N + B + l, 2 * N * B + (N + 1) * B + 3 * l
            prog.append(("G", g0, ul, ul))
            if p >= 0:
                prog.append(("G", g0 + 1, ul, p))
            if n >= 0:
                prog.append(("G", g0 + 2, ul, n))
        for b, (p, n) in enumerate(self.branches):
            for k in range(N):
                if p >= 0:
                    prog.append(("G", 2 * N * b + 2 * k, p, k))
                if n >= 0:
                    prog.append(("G", 2 * N * b + 2 * k + 1, n, k))
            if self.reactive[b]:
                if vdep[b]:
                    q = N + b
                    if p >= 0:
                        prog.append(("C", 2 * b, p, q))
                    if n >= 0:
                        prog.append(("C", 2 * b + 1, n, q))
                    prog.append(("G", 2 * N * B + (N + 1) * b, q, q))
                    for k in range(N):
                        prog.append(("G", 2 * N * B + (N + 1) * b + 1 + k, q, k))
                    prog.append(("b", 3 * b + 2, q, None))
                else:
                    for k in range(N):
                        if p >= 0:
                            prog.append(("C", 2 * B + 2 * N * b + 2 * k, p, k))
                        if n >= 0:
                            prog.append(("C", 2 * B + 2 * N * b + 2 * k + 1, n, k))
            if p >= 0:
                prog.append(("b", 3 * b, p, None))
            if n >= 0:
                prog.append(("b", 3 * b + 1, n, None))
        return prog


class _Parser:
    def __init__(self, text):
        self.toks = tokenize(text)
        self.i = 0
        self.functions = {}
        self.named = {}         # branch (a,b) name;

    def peek(self, k=0):
        return self.toks[self.i + k]

    def next(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def accept(self, val):
        if self.peek()[1] == val:
            self.i += 1
            return True
        return False

    def expect(self, val):
        t = self.next()
        if t[1] != val:
            raise VAError("expected %r, found %r" % (val, t[1]))

    def ident(self):
        t = self.next()
        if t[0] != "id":
            raise VAError("expected an identifier, found %r" % t[1])
        return t[1]

    # ---- expressions (precedence climbing) -------------------------------------------------------------------------
    def expr(self):
        c = self.or_()
        if self.accept("?"):
            a = self.expr()
            self.expect(":")
            b = self.expr()
            return ("cond", c, a, b)
        return c

    def _left(self, sub, ops):
        e = sub()
        while self.peek()[1] in ops and self.peek()[0] == "op":
            op = self.next()[1]
            e = ("bin", op, e, sub())
        return e

    def or_(self):
        return self._left(self.and_, ("||",))

    def and_(self):
        return self._left(self.cmp, ("&&",))

    def cmp(self):
        return self._left(self.add, ("==", "!=", "<", ">", "<=", ">="))

    def add(self):
        return self._left(self.mul, ("+", "-"))

    def mul(self):
        return self._left(self.unary, ("*", "/"))

    def unary(self):
        if self.accept("-"):
            return ("un", "-", self.unary())
        if self.accept("+"):
            return self.unary()
        if self.accept("!"):
            return ("un", "!", self.unary())
        return self.primary()

    def primary(self):
        kind, v = self.next()
        if kind == "num":
            if v[-1] in _SCALE and not v[-1].isdigit():
                return ("num", float(v[:-1]) * _SCALE[v[-1]])
            return ("num", float(v))
        if v == "(":
            e = self.expr()
            self.expect(")")
            return e
        if kind != "id":
            raise VAError("unexpected %r in an expression" % v)
        if v == "V" and self.peek()[1] == "(":
            self.next()
            a, b = self.probe_nets()
            return ("V", a, b)
        if v == "I" and self.peek()[1] == "(":
            raise VAError("branch current probes I(...) inside expressions are not supported")
        if v == "ddt":
            self.expect("(")
            e = self.expr()
            self.expect(")")
            return ("ddt", e)
        if v == "$limit":
            self.expect("(")
            if self.next()[1] != "V":
                raise VAError("$limit: the first argument must be a potential probe V(p[,n])")
            self.expect("(")
            a, b = self.probe_nets()
            self.expect(",")
            if self.peek()[0] == "str":
                raise VAError("$limit: the string form of the limiter is not supported; name an analog function")
            fn = self.ident()
            args = []
            while self.accept(","):
                args.append(self.expr())
            self.expect(")")
            return ("limit", a, b, fn, args, [-1])
        if v.startswith("$"):
            args = []
            if self.accept("("):
                while not self.accept(")"):
                    t = self.peek()
                    args.append(("str", self.next()[1].strip('"')) if t[0] == "str" else self.expr())
                    self.accept(",")
            if v not in ("$vt", "$temperature", "$mfactor", "$simparam"):
                raise VAError("system function %s is not supported" % v)
            return ("sys", v, args)
        if self.peek()[1] == "(":
            if v not in FUNCS and v not in self.functions:
                raise VAError("function %s is not supported" % v)
            self.next()
            args = []
            while not self.accept(")"):
                args.append(self.expr())
                self.accept(",")
            if v in self.functions:
                if len(args) != len(self.functions[v][0]):
                    raise VAError("%s takes %d argument(s)" % (v, len(self.functions[v][0])))
                return ("ucall", v, args)
            if len(args) != FUNCS[v]:
                raise VAError("%s takes %d argument(s)" % (v, FUNCS[v]))
            return ("call", v, args)
        return ("var", v)

    def probe_nets(self):
        """``a[, b])`` after ``V(`` / ``I(``; a single name may be a declared branch (``branch (a,b) name;``)."""
        a = self.ident()
        b = self.ident() if self.accept(",") else None
        self.expect(")")
        if b is None and a in self.named:
            return self.named[a]
        return a, b

    # ---- statements --------------------------------------------------------------------------------------------------
    def stmt(self):
        if self.accept("begin"):
            body = []
            while not self.accept("end"):
                body.append(self.stmt())
            return ("block", body)
        if self.accept("if"):
            self.expect("(")
            c = self.expr()
            self.expect(")")
            a = self.stmt()
            b = self.stmt() if self.accept("else") else ("block", [])
            return ("if", c, a, b)
        if self.peek()[1] == "@":
            raise VAError("event controls @(...) are not supported")
        if self.peek()[1] in ("I", "V") and self.peek(1)[1] == "(":
            acc = self.next()[1]
            self.next()
            a, b = self.probe_nets()
            if self.peek()[1] == "<+":
                self.next()
                e = self.expr()
                self.expect(";")
                if acc == "V":
                    # the one potential contribution compact models use: V(a,b) <+ 0 collapses an internal node onto its
                    # neighbour when a series resistance is zero (vasim.jl:2313-2395, 3533-3564)
                    if e != ("num", 0.0):
                        raise VAError("potential contributions other than V(a,b) <+ 0 (node collapse) are not supported")
                    return ("short", a, b)
                return ("contrib", a, b, e)
            raise VAError("expected <+ after %s(%s...)" % (acc, a))
        name = self.ident()
        if name in ("for", "while", "case", "repeat"):
            raise VAError("%s statements are not supported" % name)
        self.expect("=")
        e = self.expr()
        self.expect(";")
        return ("assign", name, e)

    # ---- module --------------------------------------------------------------------------------------------------------
    def module(self):
        self.expect("module")
        name = self.ident()
        ports = []
        if self.accept("("):
            while not self.accept(")"):
                ports.append(self.ident())
                self.accept(",")
        self.expect(";")
        nets, params, locals_, body = [], {}, [], None
        while not self.accept("endmodule"):
            t = self.peek()[1]
            if t in ("inout", "input", "output"):
                self.next()
                while not self.accept(";"):
                    self.next()
            elif t in ("electrical", "ground"):
                self.next()
                while True:
                    nets.append(self.ident())
                    if self.accept(";"):
                        break
                    self.expect(",")
            elif t == "parameter":
                self.next()
                if self.peek()[1] in ("real", "integer"):
                    self.next()
                pn = self.ident()
                self.expect("=")
                params[pn] = self.expr()
                while not self.accept(";"):       # from [..) / exclude ...: ranges are not enforced
                    self.next()
            elif t in ("real", "integer"):
                self.next()
                while True:
                    locals_.append(self.ident())
                    if self.accept(";"):
                        break
                    self.expect(",")
            elif t == "analog" and self.peek(1)[1] == "function":
                self.next(); self.next()
                if self.peek()[1] in ("real", "integer"):
                    self.next()
                fname = self.ident()
                self.expect(";")
                f_in, f_loc = [], []
                while self.peek()[1] in ("input", "real", "integer"):
                    tgt = f_in if self.next()[1] == "input" else f_loc
                    while True:
                        tgt.append(self.ident())
                        if self.accept(";"):
                            break
                        self.expect(",")
                self.functions[fname] = (f_in, [x for x in f_loc if x not in f_in], None)   # visible to its own body (recursion is refused below)
                fbody = self.stmt()
                self.expect("endfunction")
                self.functions[fname] = (f_in, [x for x in f_loc if x not in f_in], fbody[1] if fbody[0] == "block" else [fbody])
            elif t == "analog":
                self.next()
                if body is not None:
                    raise VAError("more than one analog block")
                body = self.stmt()
            elif t == "branch":
                self.next()
                self.expect("(")
                ba = self.ident()
                bb = self.ident() if self.accept(",") else None
                self.expect(")")
                while True:
                    self.named[self.ident()] = (ba, bb)
                    if self.accept(";"):
                        break
                    self.expect(",")
            else:
                raise VAError("unexpected %r in module %s" % (t, name))
        if body is None:
            raise VAError("module %s has no analog block" % name)
        for p in ports:
            if p not in nets:
                raise VAError("port %s of %s is not declared electrical" % (p, name))
        nodes = list(ports) + [x for x in nets if x not in ports]
        return VAModule(name, ports, nodes, params, locals_, body[1] if body[0] == "block" else [body], functions=dict(self.functions))


def _walk(stmts):
    for s in stmts:
        yield s
        if s[0] == "block":
            yield from _walk(s[1])
        elif s[0] == "if":
            yield from _walk([s[2], s[3]])


def _analyse(m: VAModule):
    idx = {nm: i for i, nm in enumerate(m.nodes)}

    def node(nm):
        if nm is None or nm in ("gnd", "GND"):
            return -1
        if nm not in idx:
            raise VAError("%s: net %s is not declared" % (m.name, nm))
        return idx[nm]

    names = set(m.params) | set(m.locals_)

    def check(e, names=names, in_func=False):
        k = e[0]
        if k == "var" and e[1] not in names:
            raise VAError("%s: %s is neither a parameter nor a declared variable" % (m.name, e[1]))
        if k in ("V", "ddt", "limit") and in_func:
            raise VAError("%s: %s inside an analog function" % (m.name, {"V": "V()", "ddt": "ddt()", "limit": "$limit"}[k]))
        if k == "V":
            node(e[1]); node(e[2])
        if k == "limit":
            node(e[1]); node(e[2])
            if e[3] not in m.functions:
                raise VAError("%s: $limit: unknown limiter function %s" % (m.name, e[3]))
            if len(m.functions[e[3]][0]) != 2 + len(e[4]):
                raise VAError("%s: $limit: %s takes (vnew, vold, ...) = %d arguments" % (m.name, e[3], len(m.functions[e[3]][0])))
            for a in e[4]:
                check(a, names, in_func)
            return
        for sub in e[1:]:
            if isinstance(sub, tuple):
                check(sub, names, in_func)
            elif isinstance(sub, list):
                for a in sub:
                    if isinstance(a, tuple):
                        check(a, names, in_func)

    # ---- analog functions: pure, non-recursive (a function sees the ones defined before it)
    seen = set()
    for fname, (f_in, f_loc, f_body) in m.functions.items():
        fnames = set(f_in) | set(f_loc) | {fname}

        def fcheck(e):
            if e[0] == "ucall" and e[1] not in seen:
                raise VAError("%s: %s calls %s, which is not defined before it" % (m.name, fname, e[1]))
            for sub in e[1:]:
                for a in (sub if isinstance(sub, list) else [sub]):
                    if isinstance(a, tuple) and a and isinstance(a[0], str):
                        fcheck(a)
        for s in _walk(f_body):
            if s[0] == "assign":
                if s[1] not in fnames:
                    raise VAError("%s: %s assigns %s, which it does not declare" % (m.name, fname, s[1]))
                check(s[2], fnames, True); fcheck(s[2])
            elif s[0] == "if":
                check(s[1], fnames, True); fcheck(s[1])
            elif s[0] == "contrib":
                raise VAError("%s: contribution inside the analog function %s" % (m.name, fname))
        seen.add(fname)

    # ---- V(a,b) <+ 0: collected with the conditions that guard them, which must be decidable from the parameters
    def is_static(e):
        k = e[0]
        if k in ("V", "ddt", "limit", "ucall"):
            return False
        if k == "var":
            return e[1] in m.params
        if k == "sys":
            return e[1] != "$simparam" or not (e[2] and e[2][0] == ("str", "initjct"))
        return all(is_static(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub]) if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str")

    def short_walk(stmts, guards):
        for s in stmts:
            if s[0] == "short":
                a, b = node(s[1]), node(s[2])
                if a < 0 or b < 0 or a == b:
                    raise VAError("%s: V(%s,%s) <+ 0 must join two distinct nets of the module" % (m.name, s[1], s[2]))
                for c, _ in guards:
                    if not is_static(c):
                        raise VAError("%s: V(%s,%s) <+ 0 under a condition that is not decided by the parameters" % (m.name, s[1], s[2]))
                m.shorts.append((a, b, list(guards)))
            elif s[0] == "block":
                short_walk(s[1], guards)
            elif s[0] == "if":
                short_walk([s[2]], guards + [(s[1], True)])
                short_walk([s[3]], guards + [(s[1], False)])
    short_walk(m.body, [])

    # ---- $limit call sites: numbered in source order; top level of the analog block only (vasim.jl:1278-1279)
    def sites(e, allowed):
        if e[0] == "limit":
            if not allowed:
                raise VAError("%s: $limit under a runtime conditional is unsupported" % m.name)
            br = (node(e[1]), node(e[2]))
            if br not in m.limit_branches:
                m.limit_branches.append(br)
            e[5][0] = len(m.limit_sites)
            m.limit_sites.append(m.limit_branches.index(br))
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple) and a and isinstance(a[0], str):
                    sites(a, allowed)

    def site_walk(stmts, allowed):
        for s in stmts:
            if s[0] == "assign":
                sites(s[2], allowed)
            elif s[0] == "contrib":
                sites(s[3], allowed)
            elif s[0] == "block":
                site_walk(s[1], allowed)
            elif s[0] == "if":
                sites(s[1], False)
                site_walk([s[2], s[3]], False)
    site_walk(m.body, True)

    for s in _walk(m.body):
        if s[0] == "assign":
            if s[1] not in m.locals_:
                raise VAError("%s: assignment to %s, which is not a declared variable" % (m.name, s[1]))
            check(s[2])
        elif s[0] == "contrib":
            br = (node(s[1]), node(s[2]))
            if br[0] == br[1]:
                raise VAError("%s: contribution to the degenerate branch (%s,%s)" % (m.name, s[1], s[2]))
            if br not in m.branches:
                m.branches.append(br)
            check(s[3])
        elif s[0] == "if":
            check(s[1])
    for pe in m.params.values():
        check(pe)

    # ---- which locals depend on voltages (duals), which carry a ddt() part: fixpoints over the assignments
    dual = {v: False for v in m.locals_}
    react = {v: False for v in m.locals_}

    def is_dual(e):
        k = e[0]
        if k in ("V", "limit"):
            return True
        if k == "var":
            return dual.get(e[1], False)
        return any(is_dual(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub]) if isinstance(a, tuple))

    def is_react(e):
        k = e[0]
        if k == "ddt":
            if is_react(e[1]):
                raise VAError("%s: nested ddt()" % m.name)
            return True
        if k == "var":
            return react.get(e[1], False)
        if k == "un":
            return e[1] == "-" and is_react(e[2])
        if k == "bin":
            l, r = is_react(e[2]), is_react(e[3])
            if e[1] in ("+", "-"):
                return l or r
            if e[1] == "*":
                if l and r:
                    raise VAError("%s: product of two ddt() terms" % m.name)
                return l or r
            if e[1] == "/":
                if r:
                    raise VAError("%s: division by a ddt() term" % m.name)
                return l
            if l or r:
                raise VAError("%s: ddt() inside a comparison / logical expression" % m.name)
            return False
        if k == "cond":
            if is_react(e[1]):
                raise VAError("%s: ddt() inside a condition" % m.name)
            return is_react(e[2]) or is_react(e[3])
        if k in ("call", "sys", "ucall"):
            if any(isinstance(a, tuple) and a[0] != "str" and is_react(a) for a in e[2]):
                raise VAError("%s: ddt() inside a function argument" % m.name)
        if k == "limit" and any(is_react(a) for a in e[4]):
            raise VAError("%s: ddt() inside a $limit argument" % m.name)
        return False

    changed = True
    while changed:
        changed = False
        for s in _walk(m.body):
            if s[0] == "assign":
                d, r = is_dual(s[2]) or is_react(s[2]), is_react(s[2])
                if d and not dual[s[1]]:
                    dual[s[1]] = True; changed = True
                if r and not react[s[1]]:
                    react[s[1]] = True; changed = True
    for s in _walk(m.body):
        if s[0] == "if" and is_react(s[1]):
            raise VAError("%s: ddt() inside an if condition" % m.name)
    m.var_is_dual, m.var_is_reactive = dual, react
    m.reactive = [False] * len(m.branches)
    for s in _walk(m.body):
        if s[0] == "contrib" and is_react(s[3]):
            m.reactive[m.branches.index((node(s[1]), node(s[2])))] = True
    m.is_dual, m.is_react, m.node_index = is_dual, is_react, node
    return m


def parse_module(text) -> VAModule:
    p = _Parser(text)
    m = p.module()
    if p.peek()[0] != "eof":
        raise VAError("text after endmodule (one module per source)")
    m.source = text
    return _analyse(m)
