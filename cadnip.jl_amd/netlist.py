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
  .PARAM .OPTION .TRAN .END ; continuation lines (+), comment lines (*), trailing comments (; or $)

Values are numbers with SPICE suffixes (f p n u m k meg g t), ``{name}`` / bare names of ``.PARAM``s, or names listed
in ``sweep`` -- those become ``Param`` references so that the deck can be swept on the GPU.
"""
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


def read_spice(text, models=None, includes=None, sweep=(), title=""):
    """Parse a deck.  ``models``: model / macro name -> card dict (MOSFETs: an sp_mos1 card with ``type``; diodes:
    ``is``, ``n``).  Returns ``(circuit, info)``; ``info`` holds ``options``, ``tran`` (tstep, tstop) and ``params``."""
    models = {k.lower(): v for k, v in (models or {}).items()}
    lines = _logical_lines(text, includes or {})
    info = {"options": {}, "tran": None, "params": {}}
    sweep = set(sweep)

    def val(tok):
        t = tok.strip().strip("{}'")
        if t in sweep:
            return Param(t)
        if t.lower() in info["params"]:
            return info["params"][t.lower()]
        return parse_number(t)

    sources, others = [], []
    for line in lines:
        toks = line.replace(",", " ").split()
        head = toks[0]
        hl = head.lower()
        if hl.startswith("."):
            if hl == ".end":
                break
            if hl == ".param":
                _, kv = _split_params(toks[1:])
                for k, v in kv.items():
                    info["params"][k] = val(v)
            elif hl in (".option", ".options"):
                _, kv = _split_params(toks[1:])
                info["options"].update({k: parse_number(v) for k, v in kv.items()})
            elif hl == ".tran":
                info["tran"] = tuple(parse_number(t) for t in toks[1:3])
            elif hl == ".model":
                # .model <name> nmos|pmos|d [level=1] key=value ...  (model cards, codegen.jl model registry)
                pos, kv = _split_params([t.strip("()") for t in toks[1:] if t.strip("()")])
                if len(pos) < 2:
                    raise ValueError("malformed .model card: %r" % line)
                kind_m = pos[1].lower()
                card = {k: val(v) for k, v in kv.items() if k != "level"}
                if kind_m in ("nmos", "pmos"):
                    if float(kv.get("level", "1")) != 1.0:
                        raise ValueError("only level-1 MOSFET cards (sp_mos1) have a GPU device: %r" % line)
                    card["type"] = 1 if kind_m == "nmos" else -1
                elif kind_m != "d":
                    raise ValueError("unsupported .model type %r" % pos[1])
                models[pos[0].lower()] = card
            elif hl in (".subckt", ".ends"):
                raise ValueError("hierarchical decks are not supported by this reader: flatten %r first" % line)
            continue
        kind = hl[0]
        if kind in "vi":
            p, n = toks[1], toks[2]
            rest = re.sub(r"\s*\(\s*", "(", re.sub(r"\s*\)", ")", " ".join(toks[3:])))
            wave, dc = None, 0.0
            m = re.search(r"\b(pwl|pulse|sin)\(([^)]*)\)", rest, re.I)
            if m:
                args = [parse_number(a) for a in m.group(2).split()]
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
            if rt:
                dc = val(rt[0])
            (sources if kind == "v" else others).append((kind.upper(), head, (p, n), {"dc": dc, "wave": wave}))
        elif kind in "rcl":
            others.append((kind.upper(), head, (toks[1], toks[2]), {"value": val(toks[3])}))
        elif kind in "eg":
            others.append((kind.upper(), head, tuple(toks[1:5]), {"value": val(toks[5])}))
        elif kind == "d":
            card = models.get(toks[3].lower())
            if card is None:
                raise KeyError("diode model %r is not in `models`" % toks[3])
            others.append(("D", head, (toks[1], toks[2]), {"card": card}))
        elif kind in "mx":
            pos, kv = _split_params(toks[1:])
            if len(pos) != 5:
                raise ValueError("only 4-terminal MOSFET instances / macros are supported: %r" % line)
            card = models.get(pos[4].lower())
            if card is None:
                raise KeyError("MOSFET model / macro %r is not in `models`" % pos[4])
            inst = {k: val(v) for k, v in kv.items() if k != "m"}
            others.append(("MOS1", head, tuple(pos[:4]), {"card": card, "inst": inst, "m": val(kv["m"]) if "m" in kv else 1.0}))
        elif kind == "b":
            body = line.split(None, 3)[3]
            m = re.match(r"\s*([vi])\s*=\s*(.*)$", body, re.I)
            if not m:
                raise ValueError("behavioural source needs V=expr or I=expr: %r" % line)
            expr = m.group(2).strip().strip("{}'")
            others.append(("BV" if m.group(1).lower() == "v" else "BI", head, (toks[1], toks[2]), {"expr": expr}))
        else:
            raise ValueError("unsupported element %r" % line)

    c = Circuit(title)
    for ty, name, nodes, a in sources + others:
        if ty == "V":
            c.V(name, nodes[0], nodes[1], dc=a["dc"], wave=a["wave"])
        elif ty == "I":
            c.I(name, nodes[0], nodes[1], dc=a["dc"], wave=a["wave"])
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
        elif ty == "BV":
            c.BV(name, nodes[0], nodes[1], a["expr"])
        elif ty == "BI":
            # SPICE: I=expr flows from n+ through the source to n-; BehavioralCurrentSource injects into p (devices.jl:1118-1131)
            c.BI(name, nodes[1], nodes[0], a["expr"])
    return c, info
