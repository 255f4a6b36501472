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

The constructs the reference's own model files need on top of that (models/VADistillerModels.jl/va/*.va, read by
tests/test_va_reference_models.py straight from the reference checkout)::

    `include (dropped: the standard constants are predefined), `define NAME[(args)] body, `undef, `ifdef / `ifndef / `else / `endif
    (* attributes *) in front of declarations;  aliasparam a = p;  localparam;  parameter string;  from / exclude ranges
    real x = expr, y;          // module-scope initialisers: applied at the start of every evaluation of the analog block
    analog function with output / inout arguments (by reference);  a function call as a statement
    case (e) v1, v2: stmt ... default: stmt endcase;   for (i = a; c; i = i + 1) stmt;   while (c) stmt
    $param_given(p)   $simparam(name[, default]) for every name (MNASpec fields resolve, the rest take their default;
    "iniLim" is the PCNR initjct flag, vasim.jl:1198-1206)   analysis("dc" | "static" | "tran" | ...)   $abstime
    system tasks $warning $strobe $display $write $debug $info $discontinuity $bound_step $finish (no-ops, vasim.jl:2196-2225),
    $error / $fatal (raise when a host evaluation executes them)
    white_noise / flicker_noise / noise_table contributions: zero on the DC / transient path
    (noise_enabled(::DirectStampContext) = false, value_only.jl:177)
    % ** << >> & | ^ ~   floor ceil int asin acos tan atan2 hypot asinh acosh atanh

Not supported (an error, never a silent approximation): other potential contributions ``V() <+ expr``,
``@(...)`` events other than initial_step, ``idt``, array variables, the string form of ``$limit``.
"""
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

_SCALE = {"T": 1e12, "G": 1e9, "M": 1e6, "K": 1e3, "k": 1e3, "m": 1e-3, "u": 1e-6, "n": 1e-9, "p": 1e-12, "f": 1e-15, "a": 1e-18}
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/|\(\*(?!\s*\)).*?\*\))
  | (?P<num>(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?[TGMKkmunpfa]?(?![A-Za-z_0-9]))
  | (?P<id>\$?[A-Za-z_][A-Za-z_0-9$]*)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<op><\+|==|!=|<=|>=|&&|\|\||\*\*|<<|>>|[-+*/%()<>!?:;,=@\[\]{}&|^~#.])
""", re.X | re.S)

FUNCS = {"exp": 1, "ln": 1, "log": 1, "sqrt": 1, "pow": 2, "abs": 1, "min": 2, "max": 2, "limexp": 1, "tanh": 1, "sinh": 1,
         "cosh": 1, "sin": 1, "cos": 1, "atan": 1, "tan": 1, "asin": 1, "acos": 1, "atan2": 2, "hypot": 2, "asinh": 1, "acosh": 1,
         "atanh": 1, "floor": 1, "ceil": 1, "int": 1}
NOISE_FUNCS = ("white_noise", "flicker_noise", "noise_table")
POTENTIAL_ACCESS = ("V", "Temp")        # access functions of the electrical and thermal disciplines (disciplines.vams): potential ...
FLOW_ACCESS = ("I", "Pwr")              # ... and flow
SILENT_TASKS = ("$warning", "$strobe", "$display", "$write", "$debug", "$info", "$discontinuity", "$bound_step", "$finish", "$monitor")
FATAL_TASKS = ("$error", "$fatal")
# constants.vams (Verilog-AMS LRM 2.4 annex D): the macros compact models take from the standard include
STD_DEFINES = {
    "M_E": "2.7182818284590452354", "M_LOG2E": "1.4426950408889634074", "M_LOG10E": "0.43429448190325182765", "M_LN2": "0.69314718055994530942",
    "M_LN10": "2.30258509299404568402", "M_PI": "3.14159265358979323846", "M_TWO_PI": "6.28318530717958647693", "M_PI_2": "1.57079632679489661923",
    "M_PI_4": "0.78539816339744830962", "M_1_PI": "0.31830988618379067154", "M_2_PI": "0.63661977236758134308", "M_2_SQRTPI": "1.12837916709551257390",
    "M_SQRT2": "1.41421356237309504880", "M_SQRT1_2": "0.70710678118654752440",
    "P_Q": "1.602176462e-19", "P_C": "2.99792458e8", "P_K": "1.3806503e-23", "P_H": "6.62606876e-34", "P_EPS0": "8.854187817e-12",
    "P_U0": "(4.0e-7 * 3.14159265358979323846)", "P_CELSIUS0": "273.15",
}


class VAError(ValueError):
    pass


STD_INCLUDES = ("constants.vams", "disciplines.vams", "discipline.h", "constants.h", "disciplines.h", "constants.va", "disciplines.va")


def _string_end(line, i):
    """Index of the quote that closes the string opening at line[i] (backslash escapes skipped); the last character if unterminated."""
    j = i + 1
    while j < len(line):
        if line[j] == "\\":
            j += 2
            continue
        if line[j] == '"':
            return j
        j += 1
    return len(line) - 1


def _strip_block_comments(text):
    """Block comments out, line structure kept.  A comment opener inside a string or behind // opens nothing."""
    res, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch == '"':
            j = i + 1
            while j < n and text[j] != '"' and text[j] != "\n":
                j += 2 if text[j] == "\\" else 1
            res.append(text[i:j + 1]); i = j + 1
        elif ch == "/" and text[i:i + 2] == "//":
            j = text.find("\n", i)
            j = n if j < 0 else j
            res.append(text[i:j]); i = j
        elif ch == "/" and text[i:i + 2] == "/*":
            j = text.find("*/", i + 2)
            j = n if j < 0 else j + 2
            res.append("\n" * text.count("\n", i, j)); i = j
        else:
            res.append(ch); i += 1
    return "".join(res)


def preprocess(text, defines=None, include_dir=None, _macros=None, _depth=0):
    """The compiler directives compact models use: `include (dropped -- constants.vams / disciplines.vams content is
    predefined), `define with or without arguments, `undef, `ifdef / `ifndef / `else / `elsif / `endif, and macro
    references `NAME / `NAME(args), expanded recursively.  Line continuations (backslash newline) join lines first."""
    if _macros is None:
        macros = {k: (None, v) for k, v in STD_DEFINES.items()}
        if defines:
            macros.update({k: (None, str(v)) for k, v in defines.items()})
    else:
        macros = _macros                      # an included file shares (and extends) the includer's macros
    if _depth > 20:
        raise VAError("`include nesting too deep")
    text = re.sub(r"\\\r?\n", " ", text)
    text = _strip_block_comments(text)
    out, stack = [], []          # stack of [taking, any branch taken so far]

    def expand(line, depth=0):
        if "`" not in line:
            return line
        if depth > 50:
            raise VAError("macro expansion does not terminate")
        res, i = [], 0
        while i < len(line):
            ch = line[i]
            if ch == '"':                                    # strings are opaque
                j = _string_end(line, i)
                res.append(line[i:j + 1]); i = j + 1
                continue
            if ch == "/" and line[i:i + 2] == "//":
                break
            if ch != "`":
                res.append(ch); i += 1
                continue
            m = re.match(r"`([A-Za-z_][A-Za-z_0-9]*)", line[i:])
            if not m:
                raise VAError("stray ` in %r" % line.strip())
            name = m.group(1)
            i += m.end()
            if name not in macros:
                raise VAError("macro `%s is not defined" % name)
            params, body = macros[name]
            if params is not None:
                while i < len(line) and line[i] in " \t":
                    i += 1
                if i >= len(line) or line[i] != "(":
                    raise VAError("macro `%s needs arguments" % name)
                depth_p, j, args, cur = 0, i, [], []
                while True:
                    if j >= len(line):
                        raise VAError("unterminated argument list of `%s" % name)
                    c = line[j]
                    if c == '"':                                   # a string argument may hold commas and parentheses
                        k = _string_end(line, j)
                        cur.append(line[j:k + 1]); j = k + 1
                        continue
                    if c in "([{":
                        depth_p += 1
                        if depth_p > 1:
                            cur.append(c)
                    elif c in ")]}":
                        depth_p -= 1
                        if depth_p == 0:
                            args.append("".join(cur).strip())
                            break
                        cur.append(c)
                    elif c == "," and depth_p == 1:
                        args.append("".join(cur).strip()); cur = []
                    else:
                        cur.append(c)
                    j += 1
                i = j + 1
                if len(args) != len(params) and not (len(params) == 0 and args == [""]):
                    raise VAError("macro `%s takes %d argument(s), %d given" % (name, len(params), len(args)))
                sub = body
                if params:
                    sub = re.sub(r"\b(%s)\b" % "|".join(map(re.escape, params)), lambda mm: args[params.index(mm.group(1))], body)
                res.append(expand(sub, depth + 1))
            else:
                res.append(expand(body, depth + 1))
        return "".join(res)

    for raw in text.split("\n"):
        line = raw.strip()
        taking = all(t[0] for t in stack)
        if line.startswith("`"):
            m = re.match(r"`(\w+)\s*(.*)", line)
            d, rest = m.group(1), m.group(2)
            if d in ("ifdef", "ifndef"):
                name = rest.split()[0] if rest.split() else ""
                cond = (name in macros) == (d == "ifdef")
                stack.append([taking and cond, cond])
                out.append(""); continue
            if d == "elsif":
                name = rest.split()[0] if rest.split() else ""
                outer = all(t[0] for t in stack[:-1])
                cond = (name in macros) and not stack[-1][1]
                stack[-1] = [outer and cond, stack[-1][1] or cond]
                out.append(""); continue
            if d == "else":
                outer = all(t[0] for t in stack[:-1])
                stack[-1] = [outer and not stack[-1][1], True]
                out.append(""); continue
            if d == "endif":
                stack.pop()
                out.append(""); continue
            if not taking:
                out.append(""); continue
            if d == "include":
                mi = re.match(r'"([^"]+)"', rest)
                fn = mi.group(1) if mi else ""
                if os.path.basename(fn) in STD_INCLUDES:
                    out.append(""); continue       # the standard headers: their constants are predefined (STD_DEFINES)
                if include_dir is None:
                    raise VAError("`include \"%s\": no include directory (parse the module with parse_file)" % fn)
                path = os.path.join(include_dir, fn)
                if not os.path.exists(path):
                    raise VAError("`include \"%s\": no such file in %s" % (fn, include_dir))
                out.append(preprocess(open(path).read(), None, include_dir, macros, _depth + 1))
                continue
            if d in ("timescale", "default_nettype", "resetall", "begin_keywords", "end_keywords", "pragma", "line"):
                out.append(""); continue
            if d == "define":
                m2 = re.match(r"([A-Za-z_][A-Za-z_0-9]*)(\(([^)]*)\))?\s*(.*)", rest)
                name, has_args, plist, body = m2.group(1), m2.group(2), m2.group(3), m2.group(4)
                body = re.sub(r"//.*$", "", body).strip()
                # `define NAME (x) ... : with a space before the parenthesis the parenthesis belongs to the body
                if has_args and rest[len(name):len(name) + 1] == "(":
                    params = [p.strip() for p in plist.split(",")] if plist.strip() else []
                    macros[name] = (params, body)
                else:
                    macros[name] = (None, (has_args or "") + (" " if has_args else "") + body if has_args else body)
                out.append(""); continue
            if d == "undef":
                macros.pop(rest.split()[0], None)
                out.append(""); continue
            # a macro call at the start of a line
        if not taking:
            out.append(""); continue
        out.append(expand(raw))
    if stack:
        raise VAError("unterminated `ifdef")
    return "\n".join(out)


def tokenize(text, include_dir=None, defines=None):
    text = preprocess(text, defines, include_dir)
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
    short_kind: List[str] = field(default_factory=list)               # per `shorts` entry: "named" (V(br) <+ ... at the top level of the analog block), "top"
                                                                      # (two-node, top level), "cond" (inside parameter-decided conditionals)
    short_reactive: List[bool] = field(default_factory=list)          # per `vshorts` entry: the value carries ddt() (named branches only)
    vshorts: List[int] = field(default_factory=list)                  # indices into `shorts` of the statements that are NOT a node alias: a short
                                                                      # that executes owns a branch-current unknown (see short_is_alias)
    branch_guarded: List[bool] = field(default_factory=list)          # per branch: every contribution sits under parameter-decided conditions only
    local_init: List[tuple] = field(default_factory=list)             # module-scope `real x = expr`: (name, expr) in declaration order
    aliasparams: Dict[str, str] = field(default_factory=dict)         # aliasparam alias = parameter
    param_kind: Dict[str, str] = field(default_factory=dict)          # parameter -> "real" | "integer" | "string"
    func_dirs: Dict[str, list] = field(default_factory=dict)          # analog function -> direction of every argument ("in" | "out" | "inout")
    uses_given: bool = False                                          # the module asks $param_given(...): instances carry one flag per parameter
    string_tests: List[tuple] = field(default_factory=list)           # (string parameter, literal) pairs the module compares: one host-evaluated flag each
    table_calls: List[tuple] = field(default_factory=list)            # distinct $table_model calls (inputs decided by the parameters): one host-evaluated value each
    include_dir: Optional[str] = None                                 # where `include files and $table_model tables are looked for
    hoist_vars: set = field(default_factory=set)                      # bias-independent locals computed once per parameter set (_hoist_analysis)
    cache_vars: List[str] = field(default_factory=list)               # ... those the per-call code reads: the per-device cache layout
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

    def short_is_alias(self, i):
        """The reference aliases an internal node to a *terminal* only for the pattern ``if (cond) V(int, ext) <+ 0`` at the top
        level of the analog block (detect_short_circuits, vasim.jl:2723-2818: the if-branch of a top-level conditional, one net
        internal, the other a port).  Every other executed ``V(a,b) <+ 0`` -- an else-branch as in PSP103's CollapsableR macro, a
        pair of internal nets -- is a potential contribution with its own branch current (vasim.jl:2311-2395).  ``V(a) <+ 0``
        (internal net to ground) keeps this build's alias-to-ground treatment."""
        a, b, guards = self.shorts[i][:3]
        np_ = len(self.ports)
        st = self.shorts[i][3]
        if st[3] != ("num", 0.0) or st[4] is not None:
            return False                    # a value, or a named branch: a branch with its own current (vasim.jl:3253-3280)
        if b < 0:
            return a >= np_                 # an internal net to ground is merged into ground; a terminal to ground carries a current
        return len(guards) == 1 and guards[0][1] is True and ((a >= np_) != (b >= np_))

    def aliases(self, par, given=None):
        """Node collapse of one instance: internal node index -> the node it is merged into (-1 = ground), for the
        ``V(a,b) <+ 0`` / ``V(a) <+ 0`` statements whose conditions -- decided by the parameters, possibly through
        variables the analog block computes from them -- hold with the parameter values ``par`` (``given``: the names
        of the parameters the instance sets explicitly, for $param_given)."""
        from .host_eval import collapsed_nodes
        return collapsed_nodes(self, par, given)

    def instance_structure(self, par, given=None):
        """(alias map, per `vshorts` entry: the statement executes for this instance, per branch: it is stamped) -- see
        host_eval.instance_structure."""
        from .host_eval import instance_structure
        return instance_structure(self, par, given)

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
        if self.uses_given:
            n_par += len(self.params)       # + one $param_given flag per parameter
        n_par += len(self.string_tests)     # + one flag per (string parameter == literal) test
        n_par += len(self.table_calls)      # + one value per distinct $table_model call
        NV = len(self.vshorts)
        # short currents: local unknown N + B + L + j;  G slots g_short(j) + {0: (p,I), 1: (n,I), 2: (I,p), 3: (I,n), 4 + k: (I,k)};  b slot 3B + j;
        # C slot c_short(j) = (I,I) for the named branches whose value carries ddt()
        return (N + B + L + NV, 2 * N * B + (N + 1) * B + 3 * L + (4 + N) * NV, 2 * B + 2 * N * B + sum(self.short_reactive), 3 * B + NV, n_par, 3)

    def c_short(self, j):
        N, B = self.n_nodes, len(self.branches)
        return 2 * B + 2 * N * B + sum(self.short_reactive[:j])

    def probe_short(self, e):
        """``I(br)`` / ``I(a,b)`` in an expression: the index into `vshorts` of the potential contribution whose current it reads, or
        None (a branch that carries noise only: zero on the DC / transient path)"""
        for j, si in enumerate(self.vshorts):
            st = self.shorts[si][3]
            if (e[3] is not None and st[4] == e[3]) or (e[3] is None and st[4] is None and (st[1], st[2]) == (e[1], e[2])):
                return j
        return None

    def short_current_name(self, j, instance):
        """name of the branch-current unknown of `vshorts` entry j (alloc_current! base names: vasim.jl:3262, 3274, 2336)"""
        a, b = self.shorts[self.vshorts[j]][0], self.shorts[self.vshorts[j]][1]
        kind, st = self.short_kind[self.vshorts[j]], self.shorts[self.vshorts[j]][3]
        pn = "%s_%s" % (self.nodes[a], self.nodes[b] if b >= 0 else "0")
        if kind == "named":
            return "%s_%s_I_%s" % (instance, self.name, st[4])
        return "%s_%s_I_V_%s" % (instance, self.name, pn) if kind == "top" else "%s_I_V_%s" % (instance, pn)

    def g_short(self, j):
        N, B, L = self.n_nodes, len(self.branches), len(self.limit_branches)
        return 2 * N * B + (N + 1) * B + 3 * L + (4 + N) * j

    def program(self, vdep, active=None, shorts_on=None):
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
        # executed potential contributions with a branch current.  Inside conditionals (vasim.jl:2363-2393) they are stamped where the
        # statement stands, i.e. while the body runs, before the collected branches: KCL columns, the constraint row with its partials, b.
        # At the top level of the analog block they follow the branches: named branches first (vasim.jl:3669-3746: no partials; C[I,I]
        # when the value carries ddt()), then the two-node ones (vasim.jl:3750-3815)
        def short_stamps(j, si):
            p, n = self.shorts[si][0], self.shorts[si][1]
            ui, g0 = N + B + len(self.limit_branches) + j, self.g_short(j)
            if p >= 0:
                prog.append(("G", g0, p, ui))
            if n >= 0:
                prog.append(("G", g0 + 1, n, ui))
            if p >= 0:
                prog.append(("G", g0 + 2, ui, p))
            if n >= 0:
                prog.append(("G", g0 + 3, ui, n))
            if self.short_kind[si] != "named":
                for k in range(N):
                    prog.append(("G", g0 + 4 + k, ui, k))
            prog.append(("b", 3 * B + j, ui, None))
            if self.short_reactive[j]:
                prog.append(("C", self.c_short(j), ui, ui))
        for j, si in enumerate(self.vshorts):
            if self.short_kind[si] == "cond" and shorts_on is not None and shorts_on[j]:
                short_stamps(j, si)
        for b, (p, n) in enumerate(self.branches):
            if active is not None and not active[b]:
                continue          # every contribution of the branch sits in a conditional this instance does not take (inline stamps, vasim.jl:2397-2470)
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
        for kind in ("named", "top"):
            for j, si in enumerate(self.vshorts):
                if self.short_kind[si] == kind and shorts_on is not None and shorts_on[j]:
                    short_stamps(j, si)
        return prog


class _Parser:
    def __init__(self, text, include_dir=None, defines=None):
        self.toks = tokenize(text, include_dir, defines)
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
        return self._left(self.bor, ("&&",))

    def bor(self):
        return self._left(self.bxor, ("|",))

    def bxor(self):
        return self._left(self.band, ("^",))

    def band(self):
        return self._left(self.cmp, ("&",))

    def cmp(self):
        return self._left(self.shift, ("==", "!=", "<", ">", "<=", ">="))

    def shift(self):
        return self._left(self.add, ("<<", ">>"))

    def add(self):
        return self._left(self.mul, ("+", "-"))

    def mul(self):
        return self._left(self.unary, ("*", "/", "%"))

    def unary(self):
        if self.accept("-"):
            return ("un", "-", self.unary())
        if self.accept("+"):
            return self.unary()
        if self.accept("!"):
            return ("un", "!", self.unary())
        if self.accept("~"):
            return ("un", "~", self.unary())
        return self.power()

    def power(self):
        e = self.primary()
        if self.peek() == ("op", "**"):
            self.next()
            return ("call", "pow", [e, self.unary()])       # right associative, binds tighter than unary minus on its left
        return e

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
        if kind == "str":
            return ("str", v.strip('"'))
        if kind != "id":
            raise VAError("unexpected %r in an expression" % v)
        if v == "$param_given":
            self.expect("(")
            pn = self.ident()
            self.expect(")")
            return ("given", pn)
        if v == "$port_connected":
            self.expect("(")
            self.ident()
            self.expect(")")
            return ("num", 1.0)
        if v == "analysis" and self.peek()[1] == "(":
            self.next()
            kinds = []
            while not self.accept(")"):
                t = self.next()
                if t[0] != "str":
                    raise VAError("analysis() takes string literals")
                kinds.append(t[1].strip('"'))
                self.accept(",")
            return ("analysis", kinds)
        if v in NOISE_FUNCS and self.peek()[1] == "(":
            # white_noise(pwr[, name]) / flicker_noise(pwr, exp[, name]): 0.0 on the DC / transient path; the arguments are kept for the
            # noise analysis (vasim.jl:2856-2893: the call registers a source between the enclosing contribution's nodes)
            self.next()
            args, label = [], ""
            while not self.accept(")"):
                if self.peek()[0] == "str":
                    label = self.next()[1].strip('"')
                elif v == "noise_table":
                    depth = 0
                    while depth or self.peek()[1] not in (",", ")"):      # (tables play no role here)
                        t = self.next()
                        if t[0] == "eof":
                            raise VAError("unterminated %s(" % v)
                        depth += (t[1] in "([{") - (t[1] in ")]}")
                else:
                    args.append(self.expr())
                self.accept(",")
            return ("noise", v, args, label)
        if v in POTENTIAL_ACCESS and self.peek()[1] == "(":
            self.next()
            a, b = self.probe_nets()
            return ("V", a, b)
        if v in FLOW_ACCESS and self.peek()[1] == "(":
            self.next()
            if self.peek()[1] in self.named and self.peek(1)[1] == ")":
                # the current of a NAMED branch: the branch-current unknown when the branch carries a potential contribution
                # V(br) <+ ... (vasim.jl:3632-3640), else -- parallel helper branches of correlated-noise models -- 0.0 on the DC /
                # transient path (vasim.jl:3641-3650): decided in _analyse
                br = self.next()[1]; self.next()
                a, b = self.named[br]
                return ("Iprobe", a, b, br)
            a, b = self.probe_nets()
            return ("Iprobe", a, b, None)      # the current of a two-node potential contribution, or of a branch that carries noise only (zero): _analyse
        if v == "ddt":
            self.expect("(")
            e = self.expr()
            self.expect(")")
            return ("ddt", e)
        if v == "ddx" and self.peek()[1] == "(":
            # ddx(expr, V(a)): the partial of expr with respect to the potential of net a -- the model's own small-signal
            # read-outs (gm = ddx(Ids, V(G)), ...): a plain number (its own derivatives are not tracked)
            self.next()
            e = self.expr()
            self.expect(",")
            if self.next()[1] != "V":
                raise VAError("ddx: the second argument must be a potential probe V(net)")
            self.expect("(")
            a, b = self.probe_nets()
            self.expect(")")
            return ("ddx", e, a, b)         # b: the partial with respect to a branch potential V(a,b) is (d/dV_a - d/dV_b) / 2 (vasim.jl:1168-1180)
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
            if v not in ("$vt", "$temperature", "$mfactor", "$simparam", "$abstime", "$realtime", "$table_model"):
                raise VAError("system function %s is not supported" % v)
            return ("sys", v, args)
        if self.peek()[1] == "(":
            self.next()
            args = []
            while not self.accept(")"):
                args.append(self.expr())
                self.accept(",")
            if v not in FUNCS:                                 # an analog function, possibly defined further down: checked in _analyse
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
        if self.accept(";"):
            return ("block", [])
        if self.accept("begin"):
            if self.accept(":"):
                self.ident()                                  # named block
            body = []
            while not self.accept("end"):
                if self.peek()[1] in ("real", "integer") and self.peek()[0] == "id":   # block-local declarations join the module's variables
                    self.next()
                    self.decl_list(self.block_locals, self.block_init)
                    continue
                body.append(self.stmt())
            return ("block", body)
        if self.accept("if"):
            self.expect("(")
            c = self.expr()
            self.expect(")")
            a = self.stmt()
            b = self.stmt() if self.accept("else") else ("block", [])
            return ("if", c, a, b)
        if self.accept("case"):
            self.expect("(")
            sel = self.expr()
            self.expect(")")
            items = []                                         # ([values] or None for default, statement)
            while not self.accept("endcase"):
                if self.accept("default"):
                    self.accept(":")
                    items.append((None, self.stmt()))
                    continue
                vals = [self.expr()]
                while self.accept(","):
                    vals.append(self.expr())
                self.expect(":")
                items.append((vals, self.stmt()))
            return ("case", sel, items)
        if self.accept("for"):
            self.expect("(")
            iv = self.ident(); self.expect("="); ie = self.expr(); self.expect(";")
            c = self.expr(); self.expect(";")
            sv = self.ident(); self.expect("="); se = self.expr(); self.expect(")")
            return ("for", ("assign", iv, ie), c, ("assign", sv, se), self.stmt())
        if self.accept("while"):
            self.expect("(")
            c = self.expr()
            self.expect(")")
            return ("while", c, self.stmt())
        if self.peek()[1] == "@":
            self.next()
            self.expect("(")
            ev = self.ident()
            if self.accept("("):
                while not self.accept(")"):
                    self.next()
            self.expect(")")
            body = self.stmt()
            if ev == "initial_step":
                return body                                   # runs with the first evaluation; the models use it for initialisation only
            if ev == "final_step":
                return ("block", [])
            raise VAError("event controls other than @(initial_step) / @(final_step) are not supported")
        if (self.peek()[1] in FLOW_ACCESS or self.peek()[1] in POTENTIAL_ACCESS) and self.peek(1)[1] == "(":
            acc = self.next()[1]
            self.next()
            br = self.peek()[1] if (self.peek()[1] in self.named and self.peek(1)[1] == ")") else None
            a, b = self.probe_nets()
            if self.peek()[1] == "<+":
                self.next()
                e = self.expr()
                self.expect(";")
                if acc in POTENTIAL_ACCESS:
                    # potential contribution.  V(a,b) <+ 0 collapses an internal node onto its neighbour when a series resistance is
                    # zero (vasim.jl:2313-2395, 3533-3564); with a value -- or on a named branch -- it is a branch with its own
                    # current unknown (vasim.jl:3253-3280, 3669-3815)
                    return ("short", a, b, e, br)
                return ("contrib", a, b, e)
            raise VAError("expected <+ after %s(%s...)" % (acc, a))
        kind, name = self.peek()
        if kind == "id" and name.startswith("$"):
            self.next()
            args = []
            if self.accept("("):
                depth = 1
                while depth:
                    t = self.next()
                    if t[0] == "eof":
                        raise VAError("unterminated %s(" % name)
                    depth += (t[1] == "(") - (t[1] == ")")
                    if depth:
                        args.append(t[1])
            self.expect(";")
            if name in SILENT_TASKS:
                return ("block", [])
            if name in FATAL_TASKS:
                return ("fatal", name, " ".join(args))
            raise VAError("system task %s is not supported" % name)
        name = self.ident()
        if name == "repeat":
            raise VAError("repeat statements are not supported")
        if self.peek()[1] == "(":                             # a function called for its output arguments (possibly defined further down)
            self.next()
            args = []
            while not self.accept(")"):
                args.append(self.expr())
                self.accept(",")
            self.expect(";")
            return ("callstmt", name, args)
        if self.peek()[1] == "[":
            raise VAError("array variables are not supported (%s[...])" % name)
        self.expect("=")
        e = self.expr()
        self.expect(";")
        return ("assign", name, e)

    def decl_list(self, names, inits):
        """``a, b = expr, c;`` after a type keyword"""
        while True:
            nm = self.ident()
            if self.peek()[1] == "[":
                raise VAError("array variables are not supported (%s[...])" % nm)
            names.append(nm)
            if self.accept("="):
                inits.append((nm, self.expr()))
            if self.accept(";"):
                return
            self.expect(",")

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
        inits, aliasp, pkind, fdirs = [], {}, {}, {}
        self.block_locals, self.block_init = locals_, inits
        while not self.accept("endmodule"):
            t = self.peek()[1]
            if t in ("inout", "input", "output"):
                self.next()
                while not self.accept(";"):
                    self.next()
            elif t in ("electrical", "ground", "thermal"):
                self.next()
                while True:
                    nets.append(self.ident())
                    if self.accept(";"):
                        break
                    self.expect(",")
            elif t in ("parameter", "localparam"):
                self.next()
                ptype = "real"
                if self.peek()[1] in ("real", "integer", "string"):
                    ptype = self.next()[1]
                while True:
                    pn = self.ident()
                    self.expect("=")
                    params[pn] = self.expr()
                    pkind[pn] = ptype
                    while self.peek()[1] not in (";", ",") or self._in_range():   # from [..) / exclude ...: ranges are not enforced
                        self.next()
                    if self.accept(";"):
                        break
                    self.expect(",")
            elif t == "aliasparam":
                self.next()
                al = self.ident()
                self.expect("=")
                aliasp[al] = self.ident()
                self.expect(";")
            elif t in ("real", "integer", "string", "genvar"):
                self.next()
                self.decl_list(locals_, inits)
            elif t == "analog" and self.peek(1)[1] == "function":
                self.next(); self.next()
                if self.peek()[1] in ("real", "integer"):
                    self.next()
                fname = self.ident()
                self.expect(";")
                f_args, f_dir, f_loc = [], [], []
                while self.peek()[1] in ("input", "output", "inout", "real", "integer"):
                    kw = self.next()[1]
                    while True:
                        nm = self.ident()
                        if kw in ("input", "output", "inout"):
                            f_args.append(nm); f_dir.append({"input": "in", "output": "out", "inout": "inout"}[kw])
                        else:
                            f_loc.append(nm)
                        if self.accept(";"):
                            break
                        self.expect(",")
                self.functions[fname] = (f_args, [x for x in f_loc if x not in f_args], None)   # visible to its own body (recursion is refused below)
                fdirs[fname] = f_dir
                fbody = self.stmt()
                self.expect("endfunction")
                self.functions[fname] = (f_args, [x for x in f_loc if x not in f_args], fbody[1] if fbody[0] == "block" else [fbody])
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
        seen_l = []
        for x in locals_:                                      # (a name declared twice keeps its first position)
            if x not in seen_l:
                seen_l.append(x)
        return VAModule(name, ports, nodes, params, seen_l, body[1] if body[0] == "block" else [body], functions=dict(self.functions),
                        local_init=inits, aliasparams=aliasp, param_kind=pkind, func_dirs=fdirs)

    def _in_range(self):
        """inside a ``from [a:b)`` / ``exclude ...`` clause a comma may separate range bounds? (it does not: bounds use ':'),
        so a ',' always ends the declarator; kept as a hook"""
        return False


def _walk(stmts):
    for s in stmts:
        yield s
        if s[0] == "block":
            yield from _walk(s[1])
        elif s[0] == "if":
            yield from _walk([s[2], s[3]])
        elif s[0] == "case":
            yield from _walk([it[1] for it in s[2]])
        elif s[0] == "for":
            yield from _walk([s[1], s[3], s[4]])
        elif s[0] == "while":
            yield from _walk([s[2]])


def _subexprs(s):
    """the expressions a statement evaluates itself (not those of nested statements)"""
    k = s[0]
    if k == "assign":
        return [s[2]]
    if k in ("contrib", "short"):
        return [s[3]]
    if k == "if":
        return [s[1]]
    if k == "case":
        return [s[1]] + [v for it in s[2] if it[0] is not None for v in it[0]]
    if k == "for":
        return [s[2]]
    if k == "while":
        return [s[1]]
    if k == "callstmt":
        return list(s[2])
    return []


def _analyse(m: VAModule):
    idx = {nm: i for i, nm in enumerate(m.nodes)}

    def node(nm):
        if nm is None or nm in ("gnd", "GND"):
            return -1
        if nm not in idx:
            raise VAError("%s: net %s is not declared" % (m.name, nm))
        return idx[nm]

    names = set(m.params) | set(m.locals_)
    probes = []          # branches whose current is read: a potential contribution's own current, else they must carry noise only
    # potential contributions that own a branch current whatever their value: on a named branch, or with a value other than 0
    pot_named = {s[4] for s in _walk(m.body) if s[0] == "short" and s[4] is not None}
    pot_pairs = {(node(s[1]), node(s[2])) for s in _walk(m.body) if s[0] == "short" and s[4] is None}

    def probe_is_unknown(e):
        return e[3] in pot_named if e[3] is not None else (node(e[1]), node(e[2])) in pot_pairs

    def check(e, names=names, in_func=False):
        k = e[0]
        if k in ("str", "noise", "analysis", "num"):
            return
        if k == "ddx":
            node(e[2]); node(e[3])
            check(e[1], names, in_func)
            return
        if k == "Iprobe":
            probes.append((node(e[1]), node(e[2]), e[3]))
            return
        if k == "given":
            if e[1] not in m.params and e[1] not in m.aliasparams:
                raise VAError("%s: $param_given(%s): no such parameter" % (m.name, e[1]))
            return
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

    # ---- analog functions: non-recursive; arguments are passed by reference when declared output / inout
    calls = {f: set() for f in m.functions}

    def callees(e, acc):
        if e[0] in ("ucall", "callstmt"):
            if e[1] not in m.functions:
                raise VAError("%s: call of the unknown function %s" % (m.name, e[1]))
            if len(e[2]) != len(m.functions[e[1]][0]):
                raise VAError("%s: %s takes %d argument(s), %d given" % (m.name, e[1], len(m.functions[e[1]][0]), len(e[2])))
            acc.add(e[1])
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str":
                    callees(a, acc)
    for fname, (f_in, f_loc, f_body) in m.functions.items():
        fnames = set(f_in) | set(f_loc) | {fname}
        for s in _walk(f_body):
            if s[0] == "assign" and s[1] not in fnames:
                raise VAError("%s: %s assigns %s, which it does not declare" % (m.name, fname, s[1]))
            if s[0] == "contrib":
                raise VAError("%s: contribution inside the analog function %s" % (m.name, fname))
            if s[0] == "callstmt":
                callees(s, calls[fname])
            for e in _subexprs(s):
                check(e, fnames, True); callees(e, calls[fname])
    state = {}

    def visit(f):
        if state.get(f) == 1:
            raise VAError("%s: the analog function %s is recursive" % (m.name, f))
        if state.get(f) == 2:
            return
        state[f] = 1
        for g in calls[f]:
            visit(g)
        state[f] = 2
    for f in m.functions:
        visit(f)
    m.func_order = [f for f in sorted(m.functions, key=lambda f: 0)]   # (definition order; the generators emit callees first)

    # ---- which variables are decided by the parameters alone (never assigned from a voltage, the analysis mode, the
    # initjct flag, or under a condition that depends on one)?  Flow-insensitive fixpoint.
    dyn = {v: False for v in m.locals_}

    def expr_dyn(e):
        k = e[0]
        if k in ("V", "ddt", "limit", "analysis", "ddx"):
            return True
        if k in ("given", "num", "str", "noise"):
            return False
        if k == "Iprobe":
            return probe_is_unknown(e)          # a branch-current unknown is part of the solution
        if k == "var":
            return dyn.get(e[1], False)
        if k == "sys":
            if e[1] in ("$abstime", "$realtime"):
                return True
            if e[1] == "$simparam" and e[2] and e[2][0][0] == "str" and e[2][0][1] in ("initjct", "iniLim", "iteration"):
                return True
        return any(expr_dyn(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub])
                   if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str")

    def mark(stmts, under_dyn):
        ch = False
        for st in stmts:
            k = st[0]
            if k == "assign":
                if (under_dyn or expr_dyn(st[2])) and st[1] in dyn and not dyn[st[1]]:
                    dyn[st[1]] = True; ch = True
            elif k == "callstmt" or k == "assign":
                pass
            if k in ("assign", "callstmt", "contrib", "if", "case", "while", "for"):
                for e in ([st] if k == "callstmt" else _subexprs(st)):
                    for cal in _calls_in(e):
                        d = under_dyn or any(expr_dyn(a) for a in cal[2])
                        for a, dr in zip(cal[2], m.func_dirs.get(cal[1], [])):
                            if dr != "in" and a[0] == "var" and d and a[1] in dyn and not dyn[a[1]]:
                                dyn[a[1]] = True; ch = True
            if k == "block":
                ch |= mark(st[1], under_dyn)
            elif k == "if":
                ch |= mark([st[2], st[3]], under_dyn or expr_dyn(st[1]))
            elif k == "case":
                ch |= mark([it[1] for it in st[2]], under_dyn or any(expr_dyn(e) for e in _subexprs(st)))
            elif k == "while":
                ch |= mark([st[2]], under_dyn or expr_dyn(st[1]))
            elif k == "for":
                ch |= mark([st[1], st[3], st[4]], under_dyn or expr_dyn(st[2]))
        return ch

    def _calls_in(e):
        res = []
        if e[0] in ("ucall", "callstmt"):
            res.append(e)
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str":
                    res += _calls_in(a)
        return res
    while mark(m.body, False):
        pass
    m.var_is_static = {v: not d for v, d in dyn.items()}

    # ---- V(a,b) <+ 0: collected with the conditions that guard them, which must be decidable from the parameters
    def is_static(e):
        return not expr_dyn(e)

    def _unused_is_static(e):
        k = e[0]
        if k in ("V", "ddt", "limit", "ucall", "analysis"):
            return False
        if k in ("given", "num", "str", "noise"):
            return True
        if k == "var":
            return e[1] in m.params
        if k == "sys":
            return e[1] != "$simparam" or not (e[2] and e[2][0] == ("str", "initjct"))
        return all(is_static(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub]) if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str")

    def short_walk(stmts, guards, top):
        for s in stmts:
            if s[0] == "short":
                a, b = node(s[1]), node(s[2])
                if a < 0 or a == b:
                    raise VAError("%s: V(%s,%s) <+ ... must join two distinct nets of the module" % (m.name, s[1], s[2]))
                # (V(terminal) <+ 0 ties a terminal to ground: not an alias -- it owns a branch current like any other executed short)
                for c, _ in guards:
                    if not is_static(c):
                        raise VAError("%s: V(%s,%s) <+ ... under a condition that is not decided by the parameters" % (m.name, s[1], s[2]))
                if s[4] is not None and not top:
                    raise VAError("%s: V(%s) <+ ... on a named branch inside a conditional is not supported" % (m.name, s[4]))
                m.shorts.append((a, b, list(guards), s))
                m.short_kind.append("named" if s[4] is not None else "top" if top else "cond")
            elif s[0] == "block":
                short_walk(s[1], guards, top)
            elif s[0] == "if":
                short_walk([s[2]], guards + [(s[1], True)], False)
                short_walk([s[3]], guards + [(s[1], False)], False)
            elif s[0] in ("case", "for", "while"):
                for inner in _walk([s]):
                    if inner[0] == "short":
                        raise VAError("%s: V(%s,%s) <+ ... inside a case / loop statement" % (m.name, inner[1], inner[2]))
    m.short_kind = []
    short_walk(m.body, [], True)
    # the statements that own a branch current, in the order the reference allocates the currents: named branches and two-node
    # contributions at the top level of the analog block first (branch_current_alloc, vasim.jl:3253-3280), then the ones inside
    # conditionals where they stand (vasim.jl:2366)
    rank = {"named": 0, "top": 1, "cond": 2}
    m.vshorts = sorted((i for i in range(len(m.shorts)) if not m.short_is_alias(i)), key=lambda i: (rank[m.short_kind[i]], i))
    tops = [(m.shorts[i][3][4] or (m.shorts[i][0], m.shorts[i][1])) for i in m.vshorts if m.short_kind[i] != "cond"]
    if len(set(tops)) != len(tops):
        raise VAError("%s: several potential contributions to one branch at the top level of the analog block are not supported" % m.name)


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
            elif s[0] in ("case", "for", "while", "callstmt"):
                for inner in _walk([s]):
                    for e in _subexprs(inner):
                        sites(e, False)
    site_walk(m.body, True)

    def only_noise(e):
        """a contribution whose value is noise alone adds nothing on the DC / transient path: it declares no branch"""
        return e[0] == "noise"

    for s in _walk(m.body):
        if s[0] == "assign" and s[1] not in m.locals_:
            raise VAError("%s: assignment to %s, which is not a declared variable" % (m.name, s[1]))
        if s[0] == "contrib" and not only_noise(s[3]):
            br = (node(s[1]), node(s[2]))
            if br[0] == br[1]:
                raise VAError("%s: contribution to the degenerate branch (%s,%s)" % (m.name, s[1], s[2]))
            if br not in m.branches:
                m.branches.append(br)
        if s[0] == "callstmt":
            callees(s, set())
            for a, d in zip(s[2], m.func_dirs.get(s[1], [])):
                if d != "in" and (a[0] != "var" or a[1] not in m.locals_):
                    raise VAError("%s: %s: an output argument must be a variable" % (m.name, s[1]))
        for e in _subexprs(s):
            check(e); callees(e, set())
    for pe in m.params.values():
        check(pe)
    for _, ie in m.local_init:
        check(ie)

    # ---- which locals depend on voltages (duals), which carry a ddt() part: fixpoints over the assignments
    dual = {v: False for v in m.locals_}
    react = {v: False for v in m.locals_}

    def is_dual(e):
        k = e[0]
        if k in ("V", "limit"):
            return True
        if k in ("str", "noise", "analysis", "given", "num", "Iprobe", "ddx"):
            return False
        if k == "var":
            return dual.get(e[1], False)
        return any(is_dual(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub]) if isinstance(a, tuple))

    def _react(e, R):
        """does the expression carry a ddt() part, with the variables' states in R?"""
        k = e[0]
        if k in ("str", "noise", "analysis", "given", "num", "Iprobe", "V", "ddx"):
            return False
        if k == "ddt":
            return True            # ddt of a value that already has a reactive part takes its resistive part (contrib.jl:363-369)
        if k == "var":
            return R.get(e[1], False)
        if k == "un":
            return e[1] == "-" and _react(e[2], R)
        if k == "bin":
            l, r = _react(e[2], R), _react(e[3], R)
            if e[1] in ("+", "-"):
                return l or r
            if e[1] == "*":
                return (l or r) and not (l and r)
            if e[1] == "/":
                return l and not r
            return False
        if k == "cond":
            return _react(e[2], R) or _react(e[3], R)
        # a function / comparison / $limit argument with a ddt() part is evaluated on its resistive part (the value the
        # reference's contribution dual carries in its primal slot); what is computed from it is marked (_taint) and
        # must not reach a contribution
        return False

    def is_react(e):
        return _react(e, react)

    def _taint(e, R, T):
        """does the value depend on a nonlinear function of a ddt() term (whose reactive part was dropped)?"""
        k = e[0]
        if k in ("str", "noise", "analysis", "given", "num", "V", "Iprobe", "ddx"):
            return False
        if k == "var":
            return T.get(e[1], False)
        if k == "ddt":
            return _taint(e[1], R, T)
        if k == "bin":
            l, r = _react(e[2], R), _react(e[3], R)
            if (e[1] == "*" and l and r) or (e[1] == "/" and r) or (e[1] not in ("+", "-", "*", "/") and (l or r)):
                return True
            return _taint(e[2], R, T) or _taint(e[3], R, T)
        if k == "cond":
            return _react(e[1], R) or _taint(e[1], R, T) or _taint(e[2], R, T) or _taint(e[3], R, T)
        if k in ("call", "ucall", "sys"):
            args = [a for a in e[2] if isinstance(a, tuple) and a[0] != "str"]
            return any(_react(a, R) or _taint(a, R, T) for a in args)
        if k == "limit":
            return any(_react(a, R) or _taint(a, R, T) for a in e[4])
        if k == "un":
            return (e[1] != "-" and _react(e[2], R)) or _taint(e[2], R, T)
        return False

    def taint_flow(stmts, R, T):
        """forward pass in program order: the reference's contribution duals are run-time values, so whether a variable
        carries a ddt() part is a property of the program point, not of the variable"""
        for st in stmts:
            k = st[0]
            if k == "assign":
                r, t = _react(st[2], R), _taint(st[2], R, T)
                R[st[1]], T[st[1]] = r, t
            elif k == "contrib":
                if not only_noise(st[3]) and _taint(st[3], R, T):
                    raise VAError("%s: a contribution to (%s,%s) depends on a nonlinear function of a ddt() term" % (m.name, st[1], st[2]))
            elif k == "block":
                taint_flow(st[1], R, T)
            elif k in ("if", "case"):
                arms = [st[2], st[3]] if k == "if" else [it[1] for it in st[2]] + ([] if any(it[0] is None for it in st[2]) else [("block", [])])
                outs = []
                for arm in arms:
                    R1, T1 = dict(R), dict(T)
                    taint_flow([arm], R1, T1)
                    outs.append((R1, T1))
                for v in set().union(*[o[0].keys() for o in outs]):
                    R[v] = any(o[0].get(v, False) for o in outs)
                    T[v] = any(o[1].get(v, False) for o in outs)
            elif k in ("while", "for"):
                body = [st[2]] if k == "while" else [st[1], st[4], st[3]]
                for _ in range(2):
                    R1, T1 = dict(R), dict(T)
                    taint_flow(body, R1, T1)
                    for v in R1:
                        R[v] = R.get(v, False) or R1[v]
                        T[v] = T.get(v, False) or T1[v]
            for e in ([st] if k == "callstmt" else _subexprs(st)):
                for cal in _calls_in(e):
                    bad = any(_react(a, R) or _taint(a, R, T) for a in cal[2])
                    for a, dr in zip(cal[2], m.func_dirs.get(cal[1], [])):
                        if dr != "in" and a[0] == "var":
                            R[a[1]] = False
                            T[a[1]] = bad

    def out_args(e):
        """variables written through output / inout arguments of the function calls inside ``e`` -> is any argument a dual?"""
        res = []
        if e[0] in ("ucall",) or e[0] == "callstmt":
            dirs = m.func_dirs.get(e[1], [])
            any_dual = any(is_dual(a) for a in e[2])
            for a, d in zip(e[2], dirs):
                if d != "in" and a[0] == "var":
                    res.append((a[1], any_dual))
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] not in ("str",):
                    res += out_args(a)
        return res

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
            targets = out_args(s) if s[0] == "callstmt" else [t for e in _subexprs(s) for t in out_args(e)]
            for v, d in targets:
                if d and v in dual and not dual[v]:
                    dual[v] = True; changed = True
    taint_flow(m.body, {}, {})
    # current probes: only of branches that carry nothing but noise (correlated-noise helper branches): zero on this path
    for pa, pb, pbr in probes:
        br = (pa, pb)
        if (pbr in pot_named) if pbr is not None else (br in pot_pairs):
            j = m.probe_short(("Iprobe", m.nodes[pa], m.nodes[pb] if pb >= 0 else None, pbr))
            if j is None or m.short_kind[m.vshorts[j]] == "cond":
                raise VAError("%s: I(%s) reads the current of a potential contribution inside a conditional: not supported" % (m.name, pbr or "%s,%s" % (m.nodes[pa], m.nodes[pb] if pb >= 0 else "gnd")))
            continue
        if pbr is not None:
            continue                    # a named branch without a potential contribution: its current reads 0.0 (vasim.jl:3641-3650)
        if br in m.branches:
            raise VAError("%s: I(%s,%s) is read in an expression and the branch carries a contribution: branch-current "
                          "unknowns are not supported" % (m.name, m.nodes[br[0]] if br[0] >= 0 else "gnd", m.nodes[br[1]] if br[1] >= 0 else "gnd"))
    m.var_is_dual, m.var_is_reactive = dual, react
    # ---- branches whose every contribution stands under parameter-decided conditions: the reference stamps such contributions
    # inline, where and when the statement executes (vasim.jl:2270-2470), so an instance that never takes them has no stamps
    guarded = [True] * len(m.branches)

    def contrib_walk(stmts, under):          # under: "top" (no enclosing conditional), "static" (parameter-decided ones only), "dyn"
        for s in stmts:
            if s[0] == "contrib":
                if s[3][0] != "noise" and under != "static":
                    guarded[m.branches.index((node(s[1]), node(s[2])))] = False
            elif s[0] == "block":
                contrib_walk(s[1], under)
            elif s[0] == "if":
                contrib_walk([s[2], s[3]], "dyn" if (under == "dyn" or not is_static(s[1])) else "static")
            elif s[0] in ("case", "for", "while"):
                contrib_walk([x for x in _walk([s]) if x[0] == "contrib"], "dyn")
    contrib_walk(m.body, "top")
    m.branch_guarded = guarded
    m.reactive = [False] * len(m.branches)
    for s in _walk(m.body):
        if s[0] == "contrib" and not only_noise(s[3]) and is_react(s[3]):
            m.reactive[m.branches.index((node(s[1]), node(s[2])))] = True
    m.short_reactive = []
    for i in m.vshorts:
        st = m.shorts[i][3]
        r = is_react(st[3])
        if r and m.short_kind[i] != "named":
            raise VAError("%s: V(%s,%s) <+ ... with a ddt() term is supported on named branches only" % (m.name, st[1], st[2]))
        m.short_reactive.append(r)
    for i in range(len(m.shorts)):
        if i not in m.vshorts and m.shorts[i][3][3] != ("num", 0.0):
            raise VAError("%s: internal: a potential contribution with a value was classified as a node alias" % m.name)

    def has_given(e):
        if e[0] == "given":
            return True
        return any(has_given(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub])
                   if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str")
    def string_tests(e):
        if e[0] == "bin" and e[1] in ("==", "!="):
            for x, y in ((e[2], e[3]), (e[3], e[2])):
                if x[0] == "var" and m.param_kind.get(x[1]) == "string" and y[0] == "str" and (x[1], y[1]) not in m.string_tests:
                    m.string_tests.append((x[1], y[1]))
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str":
                    string_tests(a)
    def table_calls(e):
        if e[0] == "sys" and e[1] == "$table_model":
            if len(e[2]) < 3 or e[2][-1][0] != "str" or e[2][-2][0] != "str":
                raise VAError("%s: $table_model(inputs..., \"file\", \"control\")" % m.name)
            for a in e[2][:-2]:
                if not _params_only(a):
                    raise VAError("%s: $table_model of a quantity that is not decided by the parameters alone is not supported" % m.name)
            if e not in m.table_calls:
                m.table_calls.append(e)
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple) and a and isinstance(a[0], str) and a[0] != "str":
                    table_calls(a)

    def _params_only(e):
        if e[0] == "num":
            return True
        if e[0] == "var":
            return e[1] in m.params
        if e[0] in ("un", "bin", "cond", "call"):
            return all(_params_only(a) for sub in e[1:] for a in (sub if isinstance(sub, list) else [sub]) if isinstance(a, tuple))
        return False
    bodies = [m.body] + [f[2] for f in m.functions.values()]
    for body in bodies:
        for st in _walk(body):
            for e in ([st] if st[0] == "callstmt" else _subexprs(st)):
                string_tests(e)
                table_calls(e)
    m.uses_given = any(has_given(e) for body in bodies for st in _walk(body) for e in ([st] if st[0] == "callstmt" else _subexprs(st))) \
        or any(has_given(ie) for _, ie in m.local_init)
    m.is_dual, m.is_react, m.node_index = is_dual, is_react, node
    m.expr_is_static = lambda e: not expr_dyn(e)
    _hoist_analysis(m)
    return m


def _hoist_analysis(m):
    """Which bias-independent locals can be computed once per parameter set instead of once per stamp call (the reference re-evaluates the
    whole analog block per call; OSDI compilers split model / instance setup from eval the same way).
    ``m.hoist_vars``: static locals whose every write precedes, in program order, every read -- their value at any read is their final
    value -- and that are written neither in a loop nor through an output argument.  ``m.cache_vars``: those of them that anything remaining
    in the per-call code reads (ordered: the layout of the per-device cache).  Temporaries that are reused (read before a later write)
    stay in the per-call code."""
    pos = [0]
    first_read, last_write, barred = {}, {}, set()

    def reads(e, acc):
        if not isinstance(e, tuple) or not e or e[0] == "noise":      # (the device code never evaluates a noise power)
            return
        if e[0] == "var":
            acc.add(e[1])
        for sub in e[1:]:
            for a in (sub if isinstance(sub, list) else [sub]):
                if isinstance(a, tuple):
                    reads(a, acc)

    def note_reads(exprs, p):
        acc = set()
        for e in exprs:
            reads(e, acc)
        for v in acc:
            first_read.setdefault(v, p)

    def walk(stmts, loop):
        for s in stmts:
            pos[0] += 1
            p, k = pos[0], s[0]
            if k == "assign":
                note_reads([s[2]], p)
                last_write[s[1]] = p
                if loop:
                    barred.add(s[1])
            elif k in ("contrib", "short"):
                note_reads([s[3]], p)
            elif k == "block":
                walk(s[1], loop)
            elif k == "if":
                note_reads([s[1]], p)
                walk([s[2], s[3]], loop)
            elif k == "case":
                note_reads([s[1]] + [v for vals, _ in s[2] if vals is not None for v in vals], p)
                walk([body for _, body in s[2]], loop)
            elif k == "while":
                note_reads([s[1]], p)
                walk([s[2]], True)
            elif k == "for":
                walk([s[1]], True)
                note_reads([s[2]], p)
                walk([s[4], s[3]], True)
            elif k == "callstmt":
                note_reads(s[2], p)
                for a, d in zip(s[2], m.func_dirs.get(s[1]) or []):
                    if d != "in" and a[0] == "var":
                        barred.add(a[1])
    for nm, ie in m.local_init:
        pos[0] += 1
        note_reads([ie], pos[0])
        last_write[nm] = pos[0]
    walk(m.body, False)
    H = {v for v in m.locals_ if m.var_is_static.get(v) and v in last_write and v not in barred
         and (v not in first_read or last_write[v] < first_read[v])}
    # a hoisted variable may only be computed from parameters, system constants and other hoisted variables ... or from static
    # temporaries, which the setup pass computes as well (it runs every static statement): nothing to check.  What the per-call code
    # still reads:
    used = set()

    def stay(stmts):
        for s in stmts:
            k = s[0]
            if k == "assign":
                if s[1] not in H:
                    reads(s[2], used)
            elif k in ("contrib", "short"):
                reads(s[3], used)
            elif k == "block":
                stay(s[1])
            elif k == "if":
                reads(s[1], used)
                stay([s[2], s[3]])
            elif k == "case":
                reads(s[1], used)
                for vals, body in s[2]:
                    for v in (vals or []):
                        reads(v, used)
                    stay([body])
            elif k == "while":
                reads(s[1], used); stay([s[2]])
            elif k == "for":
                stay([s[1]]); reads(s[2], used); stay([s[4], s[3]])
            elif k == "callstmt":
                for a in s[2]:
                    reads(a, used)
    stay(m.body)
    m.hoist_vars = H
    m.cache_vars = [v for v in m.locals_ if v in H and v in used]


def parse_file(path, defines=None) -> VAModule:
    """parse_module on a file: `include directives resolve relative to the file's directory"""
    return parse_module(open(path).read(), os.path.dirname(os.path.abspath(path)), defines)


def parse_module(text, include_dir=None, defines=None) -> VAModule:
    p = _Parser(text, include_dir, defines)
    m = p.module()
    if p.peek()[0] != "eof":
        raise VAError("text after endmodule (one module per source)")
    m.source = text
    m.include_dir = include_dir
    return _analyse(m)
