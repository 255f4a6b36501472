"""SPICE deck -> device table (the step in front of the hot path).

The reference turns a deck into a Julia builder through its parser, semantic analysis and code generator
(/root/reference/src/spc/sema.jl:630-820, src/spc/codegen.jl:3437-3518).  The GPU path only needs the flattened
result: one ``Circuit`` row per instance, voltage sources first (codegen.jl:3130-3149), so that decks such as the
reference's own ``test/DFF/DFF_cap_all.cir`` feed it without hand transcription.  This reader covers the flat-deck
subset those tests use:

  R C L            value
  V I              DC value and / or PWL(...) PULSE(...) SIN(...)
  E G              linear VCVS / VCCS
  D                model -> (is, n)
  M  / X           4-terminal MOSFET: ``Mname d g s b model W= L= [M=]``; the gf180 decks instantiate the PDK's
                   device macros as ``Xname d g s b nfet_06v0 W= L=`` -- resolved against ``models`` the same way
  B                behavioural source  V=expr / I=expr  (bsource.py)
  .INCLUDE .LIB    ``includes``: name -> text (a .LIB that is not supplied is skipped: model cards come from ``models``)
  .MODEL           nmos / pmos (level 1 -> sp_mos1 card) and d cards; cards may also be passed in ``models``
  X                ``Xname nets... subckt [k=v]``: a call of a ``.SUBCKT`` defined in the deck (or an include) is expanded;
                   ``Xname nets... module [k=v]``: an instance of a Verilog-A module compiled into the library (va/)
  .SUBCKT .ENDS    hierarchical definitions, flattened with the reference's ``prefix_name`` naming; .GLOBAL nets
  .PARAM .OPTION .TRAN .END ; continuation lines (+), comment lines (*), trailing comments (; or $)

Values are numbers with SPICE suffixes (f p n u m k meg g t), ``{expr}`` / ``'expr'`` expressions over ``.PARAM``s and
subcircuit parameters (+ - * / ** and one-line functions), or names listed in ``sweep`` -- those become ``Param``
references (they may pass affinely through expressions) so that the deck can be swept on the GPU.
"""
import ast
import math
import re

from .circuit import Circuit, Param

_SUFFIX = {"t": 12, "g": 9, "meg": 6, "k": 3, "m": -3, "u": -6, "n": -9, "p": -12, "f": -15, "a": -18}
_NUM = re.compile(r"^([+-]?(?:\d+\.?\d*|\.\d+))(?:e([+-]?\d+))?(meg|mil|[tgkmunpfa])?[a-z]*$")


def parse_number(tok):
    """SPICE number with scale suffix; the suffix is folded into the decimal exponent so that ``100n`` is the double
    nearest to 1e-7 (a multiplication by 1e-9 would be one ulp off)."""
    m = _NUM.match(tok.strip().lower())
    if not m:
        raise ValueError("not a SPICE number: %r" % tok)
    mant, exp, suf = m.group(1), int(m.group(2) or 0), m.group(3)
    if suf == "mil":
        return float("%se%d" % (mant, exp)) * 25.4e-6
    return float("%se%d" % (mant, exp + (_SUFFIX[suf] if suf else 0)))


def _logical_lines(text, includes):
    """Comment-free logical lines with continuations joined and includes expanded."""
    out = []
    for raw in text.splitlines():
        line = re.split(r"\s;|\s\$", " " + raw, 1)[0].strip()
        if not line or line.startswith("*"):
            continue
        if line.startswith("+"):
            if not out:
                raise ValueError("continuation line without a predecessor")
            out[-1] += " " + line[1:].strip()
            continue
        out.append(line)
    res = []
    for line in out:
        # an expression is one token: drop the blanks inside {...} and '...'
        line = re.sub(r"\{[^}]*\}|'[^']*'", lambda m: re.sub(r"\s+", "", m.group(0)), line)
        head = line.split()[0].lower()
        if head in (".include", ".inc", ".lib"):
            name = line.split(None, 1)[1].strip().split()[0].strip("\"'")
            key = next((k for k in includes if k == name or name.endswith("/" + k) or name.endswith(k)), None)
            if key is not None:
                res.extend(_logical_lines(includes[key], includes))
            elif head != ".lib":
                raise FileNotFoundError("included file %r was not supplied" % name)
            continue
        res.append(line)
    return res


def _split_params(tokens):
    """Positional tokens and name=value pairs (``a = b`` spellings are joined first)."""
    s = re.sub(r"\s*=\s*", "=", " ".join(tokens))
    pos, kv = [], {}
    for t in s.split():
        if "=" in t:
            k, v = t.split("=", 1)
            kv[k.lower()] = v
        else:
            pos.append(t)
    return pos, kv


class _Affine:
    """value = scale * sweep[name] + offset: what a ``Param`` can carry through +, -, * const, / const."""

    def __init__(self, name, scale=1.0, offset=0.0):
        self.name, self.scale, self.offset = name, scale, offset

    @staticmethod
    def of(x):
        if isinstance(x, Param):
            return _Affine(x.name, x.scale, x.offset)
        return x

    def to_param(self):
        return Param(self.name, scale=self.scale, offset=self.offset)


def _arith(op, a, b):
    import operator
    fa, fb = isinstance(a, _Affine), isinstance(b, _Affine)
    if not fa and not fb:
        return {"+": operator.add, "-": operator.sub, "*": operator.mul, "/": operator.truediv, "**": operator.pow}[op](a, b)
    if op in "+-":
        sgn = 1.0 if op == "+" else -1.0
        if fa and fb:
            if a.name != b.name:
                raise ValueError("expression mixes the sweep parameters %r and %r" % (a.name, b.name))
            return _Affine(a.name, a.scale + sgn * b.scale, a.offset + sgn * b.offset)
        if fa:
            return _Affine(a.name, a.scale, a.offset + sgn * b)
        return _Affine(b.name, sgn * b.scale, a + sgn * b.offset)
    if op == "*" and not (fa and fb):
        k, x = (b, a) if fa else (a, b)
        return _Affine(x.name, x.scale * k, x.offset * k)
    if op == "/" and fa and not fb:
        return _Affine(a.name, a.scale / b, a.offset / b)
    raise ValueError("expression is not affine in the sweep parameter")


_FUNCS = {"sqrt": math.sqrt, "exp": math.exp, "ln": math.log, "log": math.log, "log10": math.log10, "abs": abs, "min": min, "max": max,
          "pow": pow, "sin": math.sin, "cos": math.cos, "tan": math.tan, "atan": math.atan, "floor": math.floor, "ceil": math.ceil,
          "int": lambda x: float(int(x))}


def eval_expr(text, lookup):
    """``{...}`` / ``'...'`` parameter expression: numbers with SPICE suffixes, names (``lookup(name)`` -> float or Param),
    + - * / ** (also ^), parentheses and the usual one-line functions.  A sweep parameter may appear affinely."""
    src = re.sub(r"(?<![\w.])((?:\d+\.?\d*|\.\d+)(?:e[+-]?\d+)?(?:meg|mil|[tgkmunpfa])?)(?![\w.])",
                 lambda m: repr(parse_number(m.group(1))), text.strip().lower().replace("^", "**"))
    try:
        tree = ast.parse(src, mode="eval")
    except SyntaxError as e:
        raise ValueError("cannot parse expression %r" % text) from e

    def ev(nd):
        if isinstance(nd, ast.Expression):
            return ev(nd.body)
        if isinstance(nd, ast.Constant) and isinstance(nd.value, (int, float)):
            return float(nd.value)
        if isinstance(nd, ast.Name):
            return _Affine.of(lookup(nd.id))
        if isinstance(nd, ast.UnaryOp) and isinstance(nd.op, (ast.USub, ast.UAdd)):
            v = ev(nd.operand)
            return v if isinstance(nd.op, ast.UAdd) else _arith("*", -1.0, v)
        if isinstance(nd, ast.BinOp):
            op = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/", ast.Pow: "**"}.get(type(nd.op))
            if op is None:
                raise ValueError("unsupported operator in %r" % text)
            return _arith(op, ev(nd.left), ev(nd.right))
        if isinstance(nd, ast.Call) and isinstance(nd.func, ast.Name) and nd.func.id in _FUNCS:
            args = [ev(a) for a in nd.args]
            if any(isinstance(a, _Affine) for a in args):
                raise ValueError("function of a sweep parameter in %r" % text)
            return float(_FUNCS[nd.func.id](*args))
        raise ValueError("unsupported construct in expression %r" % text)

    v = ev(tree)
    return v.to_param() if isinstance(v, _Affine) else float(v)


_GROUND = ("0", "gnd", "gnd!")


def read_spice(text, models=None, includes=None, sweep=(), title=""):
    """Parse a deck.  ``models``: model / macro name -> card dict (MOSFETs: an sp_mos1 card with ``type``; diodes:
    ``is``, ``n``).  Returns ``(circuit, info)``; ``info`` holds ``options``, ``tran`` (tstep, tstop) and ``params``.

    Hierarchy: ``.SUBCKT name ports... [PARAMS:] k=v`` ... ``.ENDS`` definitions are expanded at every ``X`` call into
    the flat device table, with the reference's naming: instance ``m1`` inside ``x1`` inside ``xu1`` becomes
    ``xu1_x1_m1``, an internal net ``n`` becomes ``xu1_x1_n`` (codegen.jl:745-757); ports, ground and ``.GLOBAL`` nets
    keep their outer names.  Call parameters are evaluated in the caller's scope and override the definition's
    defaults; a body sees its own parameters first, then the enclosing scopes."""
    models = {k.lower(): v for k, v in (models or {}).items()}
    lines = _logical_lines(text, includes or {})
    info = {"options": {}, "tran": None, "params": {}}
    sweep = set(sweep)
    globals_ = set()
    subckts = {}

    # ---- pass 1: lift .SUBCKT bodies out of the line list
    top, stack = [], []
    for line in lines:
        hl = line.split()[0].lower()
        if hl == ".subckt":
            toks = line.replace(",", " ").split()
            pos, kv = _split_params([t for t in toks[2:] if t.lower() != "params:"])
            stack.append({"name": toks[1].lower(), "ports": pos, "defaults": kv, "body": []})
        elif hl == ".ends":
            if not stack:
                raise ValueError(".ENDS without .SUBCKT")
            sc = stack.pop()
            if stack:
                raise ValueError("nested .SUBCKT definitions are not supported (%s inside %s)" % (sc["name"], stack[-1]["name"]))
            subckts[sc["name"]] = sc
        elif stack:
            stack[-1]["body"].append(line)
        else:
            top.append(line)
    if stack:
        raise ValueError(".SUBCKT %s is not closed by .ENDS" % stack[-1]["name"])

    sources, others = [], []
    from . import va
    va_names = va.module_names()          # lower-case name -> module name; a module is parsed when a card or an instance uses it

    class Scope:
        def __init__(self, prefix="", nmap=None, params=None, parent=None):
            self.prefix, self.nmap, self.params, self.parent = prefix, nmap or {}, params if params is not None else {}, parent

        def lookup(self, name):
            sc = self
            while sc is not None:
                if name in sc.params:
                    return sc.params[name]
                sc = sc.parent
            if name in sweep_l:
                return Param(sweep_l[name])
            raise KeyError("unknown parameter %r" % name)

        def val(self, tok):
            t = tok.strip()
            if t in sweep:
                return Param(t)
            if t[:1] in "{'" and t[-1:] in "}'":
                return eval_expr(t[1:-1], self.lookup)
            try:
                return parse_number(t)
            except ValueError:
                return eval_expr(t, self.lookup)

        def node(self, n):
            if n.lower() in _GROUND or n.lower() in globals_:
                return n
            if n in self.nmap:
                return self.nmap[n]
            return self.prefix + "_" + n if self.prefix else n

        def name(self, n):
            return self.prefix + "_" + n if self.prefix else n

    sweep_l = {k.lower(): k for k in sweep}

    def bexpr(expr, sc):
        """Rename the nets inside V(...) and fold parameter names into numbers."""
        def vsub(m):
            return "V(" + ",".join(sc.node(a.strip()) for a in m.group(1).split(",")) + ")"
        out = re.sub(r"\b[vV]\(([^)]*)\)", vsub, expr)
        if sc.prefix or sc.params or info["params"]:
            def psub(m):
                w = m.group(0)
                if m.end() < len(out) and out[m.end():m.end() + 1] == "(":
                    return w
                try:
                    v = sc.lookup(w.lower())
                except KeyError:
                    return w
                if isinstance(v, Param):
                    raise ValueError("behavioural source uses the sweep parameter %r" % w)
                return repr(float(v))
            parts = re.split(r"(V\([^)]*\))", out)
            out = "".join(p if p.startswith("V(") else re.sub(r"(?<![\w.])[A-Za-z_]\w*", psub, p) for p in parts)
        return out

    def handle(line, sc, depth):
        toks = line.replace(",", " ").split()
        head = toks[0]
        hl = head.lower()
        if hl.startswith("."):
            if hl == ".param":
                _, kv = _split_params(toks[1:])
                for k, v in kv.items():
                    sc.params[k] = sc.val(v)
            elif sc.prefix:
                if hl not in (".model",):
                    raise ValueError("%s is not allowed inside a .SUBCKT body" % head)
            elif hl in (".option", ".options"):
                _, kv = _split_params(toks[1:])
                info["options"].update({k: parse_number(v) for k, v in kv.items()})
            elif hl == ".tran":
                info["tran"] = tuple(parse_number(t) for t in toks[1:3])
            elif hl == ".global":
                globals_.update(t.lower() for t in toks[1:])
            if hl == ".model":
                # .model <name> nmos|pmos|d [level=1] key=value ...  (model cards, codegen.jl model registry)
                pos, kv = _split_params([t.strip("()") for t in toks[1:] if t.strip("()")])
                if len(pos) < 2:
                    raise ValueError("malformed .model card: %r" % line)
                kind_m = pos[1].lower()
                card = {k: sc.val(v) for k, v in kv.items() if k != "level"}
                if kind_m in ("nmos", "pmos"):
                    if float(kv.get("level", "1")) != 1.0:
                        raise ValueError("only level-1 MOSFET cards (sp_mos1) have a GPU device: %r" % line)
                    card["type"] = 1 if kind_m == "nmos" else -1
                elif kind_m in va_names:
                    # .model <name> <verilog-a module> k=v ...: a card of a generated module (test/mna/psp103_integration.jl:44:
                    # ".model nch psp103va type=1"); instances merge their own parameters over it
                    card["__va__"] = va_names[kind_m]
                elif kind_m != "d":
                    raise ValueError("unsupported .model type %r" % pos[1])
                models[pos[0].lower()] = card
            return
        kind = hl[0]
        name = sc.name(head)
        if kind in "vi":
            p, n = sc.node(toks[1]), sc.node(toks[2])
            rest = re.sub(r"\s*\(\s*", "(", re.sub(r"\s*\)", ")", " ".join(toks[3:])))
            wave, dc = None, 0.0
            m = re.search(r"\b(pwl|pulse|sin)\(([^)]*)\)", rest, re.I)
            if not m:      # the same functions without parentheses: the rest of the card is the argument list
                m = re.search(r"\b(pwl|pulse|sin)\s+(.*)$", rest, re.I)
            if m:
                args = [sc.val(a) for a in m.group(2).split()]
                if any(isinstance(a, Param) for a in args):
                    raise ValueError("waveform arguments cannot be swept (scale the source instead): %r" % line)
                fn = m.group(1).lower()
                if fn == "pwl":
                    wave = ("pwl", args[0::2], args[1::2])
                    dc = args[1]
                elif fn == "pulse":
                    args += [0.0] * (7 - len(args))
                    wave = ("pulse",) + tuple(args[:7])
                    dc = args[0]
                else:
                    wave = ("sin",) + tuple(args[:6])
                    dc = args[0]
                rest = rest[:m.start()] + rest[m.end():]
            rt = [t for t in rest.split() if t.lower() != "dc"]
            ac = 0.0
            low = [t.lower() for t in rt]
            if "ac" in low:      # AC mag [phase in degrees]  (test/ac.jl:21, 101-108)
                k = low.index("ac")
                vals = []
                for t in rt[k + 1:k + 3]:
                    try:
                        vals.append(float(sc.val(t)))
                    except (ValueError, KeyError):
                        break
                mag = vals[0] if vals else 1.0
                ac = mag * complex(math.cos(math.radians(vals[1])), math.sin(math.radians(vals[1]))) if len(vals) > 1 else mag
                rt = rt[:k] + rt[k + 1 + len(vals):]
            if rt:
                dc = sc.val(rt[0])
            (sources if kind == "v" else others).append((kind.upper(), name, (p, n), {"dc": dc, "wave": wave, "ac": ac}))
        elif kind in "rcl":
            others.append((kind.upper(), name, (sc.node(toks[1]), sc.node(toks[2])), {"value": sc.val(toks[3])}))
        elif kind in "eg":
            others.append((kind.upper(), name, tuple(sc.node(t) for t in toks[1:5]), {"value": sc.val(toks[5])}))
        elif kind == "d":
            card = models.get(toks[3].lower())
            if card is None:
                raise KeyError("diode model %r is not in `models`" % toks[3])
            others.append(("D", name, (sc.node(toks[1]), sc.node(toks[2])), {"card": card}))
        elif kind in "mxn":
            pos, kv = _split_params(toks[1:])
            va_card = models.get(pos[-1].lower()) if pos else None
            if kind in "mn" and isinstance(va_card, dict) and "__va__" in va_card:
                # Mname / Nname nets... card [k=v]: instance of a Verilog-A module through a .model card (N: the OSDI element letter
                # of ngspice decks, benchmarks/vacask/ring/cedarsim/models.inc:4)
                mod = va.get(va_card["__va__"])[1]
                if len(pos) - 1 != len(mod.ports):
                    raise ValueError("%s: %d nets for the %d ports of %s" % (head, len(pos) - 1, len(mod.ports), mod.name))
                inst = {k: v for k, v in va_card.items() if k != "__va__"}
                inst.update({k: sc.val(v) for k, v in kv.items() if k != "m"})
                others.append(("VA", name, tuple(sc.node(t) for t in pos[:-1]), {"module": mod.name, "inst": inst, "m": sc.val(kv["m"]) if "m" in kv else 1.0}))
                return
            if kind == "n":
                raise ValueError("N element without a Verilog-A model card: %r" % line)
            sub = subckts.get(pos[-1].lower()) if kind == "x" and pos else None
            if sub is not None:
                if depth > 32:
                    raise ValueError("subcircuit recursion through %s" % sub["name"])
                if len(pos) - 1 != len(sub["ports"]):
                    raise ValueError("%s: %d nets for the %d ports of %s" % (head, len(pos) - 1, len(sub["ports"]), sub["name"]))
                unknown = [k for k in kv if k not in sub["defaults"] and k != "m"]
                if unknown:
                    raise ValueError("%s: %s has no parameter %s" % (head, sub["name"], ", ".join(unknown)))
                pars = {k: sc.val(v) for k, v in kv.items()}                 # call values: the caller's scope
                inner = Scope(name, dict(zip(sub["ports"], (sc.node(t) for t in pos[:-1]))), pars, sc)
                for k, v in sub["defaults"].items():                         # defaults may refer to earlier parameters
                    if k not in pars:
                        pars[k] = inner.val(v)
                for body_line in sub["body"]:
                    handle(body_line, inner, depth + 1)
                return
            if kind == "x" and pos and pos[-1].lower() in va_names:
                # instance of a Verilog-A module that is compiled into the library (cadnip.jl_amd/va)
                mod = va.get(va_names[pos[-1].lower()])[1]
                if len(pos) - 1 != len(mod.ports):
                    raise ValueError("%s: %d nets for the %d ports of %s" % (head, len(pos) - 1, len(mod.ports), mod.name))
                inst = {k: sc.val(v) for k, v in kv.items() if k != "m"}
                others.append(("VA", name, tuple(sc.node(t) for t in pos[:-1]), {"module": mod.name, "inst": inst, "m": sc.val(kv["m"]) if "m" in kv else 1.0}))
                return
            if len(pos) != 5:
                raise ValueError("only 4-terminal MOSFET instances / macros, Verilog-A modules and .SUBCKT calls are supported: %r" % line)
            card = models.get(pos[4].lower())
            if card is None:
                raise KeyError("MOSFET model / macro / subcircuit %r is not defined" % pos[4])
            inst = {k: sc.val(v) for k, v in kv.items() if k != "m"}
            others.append(("MOS1", name, tuple(sc.node(t) for t in pos[:4]), {"card": card, "inst": inst, "m": sc.val(kv["m"]) if "m" in kv else 1.0}))
        elif kind == "b":
            body = line.split(None, 3)[3]
            m = re.match(r"\s*([vi])\s*=\s*(.*)$", body, re.I)
            if not m:
                raise ValueError("behavioural source needs V=expr or I=expr: %r" % line)
            expr = bexpr(m.group(2).strip().strip("{}'"), sc)
            others.append(("BV" if m.group(1).lower() == "v" else "BI", name, (sc.node(toks[1]), sc.node(toks[2])), {"expr": expr}))
        else:
            raise ValueError("unsupported element %r" % line)

    root = Scope("", {}, info["params"], None)
    for line in top:
        if line.split()[0].lower() == ".end":
            break
        handle(line, root, 0)

    c = Circuit(title)
    for ty, name, nodes, a in sources + others:
        if ty == "V":
            c.V(name, nodes[0], nodes[1], dc=a["dc"], wave=a["wave"], ac=a.get("ac", 0.0))
        elif ty == "I":
            c.I(name, nodes[0], nodes[1], dc=a["dc"], wave=a["wave"], ac=a.get("ac", 0.0))
        elif ty == "R":
            c.R(name, nodes[0], nodes[1], a["value"])
        elif ty == "C":
            c.C(name, nodes[0], nodes[1], a["value"])
        elif ty == "L":
            c.L(name, nodes[0], nodes[1], a["value"])
        elif ty == "E":
            c.E(name, nodes[0], nodes[1], nodes[2], nodes[3], a["value"])
        elif ty == "G":
            c.G(name, nodes[0], nodes[1], nodes[2], nodes[3], a["value"])
        elif ty == "D":
            c.D(name, nodes[0], nodes[1], Is=a["card"].get("is", 1e-14), n_=a["card"].get("n", 1.0))
        elif ty == "MOS1":
            c.MOS1(name, nodes[0], nodes[1], nodes[2], nodes[3], a["card"], m=a["m"], **a["inst"])
        elif ty == "VA":
            c.VA(name, a["module"], nodes, m=a["m"], **a["inst"])
        elif ty == "BV":
            c.BV(name, nodes[0], nodes[1], a["expr"])
        elif ty == "BI":
            # SPICE: I=expr flows from n+ through the source to n-; BehavioralCurrentSource injects into p (devices.jl:1118-1131)
            c.BI(name, nodes[1], nodes[0], a["expr"])
    return c, info
