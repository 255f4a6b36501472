"""Behavioural-source expressions -> postfix programs (include/cadnip_hip.h: CadnipBsrcOp).

The reference's ``BehavioralVoltageSource`` / ``BehavioralCurrentSource`` carry a Julia closure
``value_fn(get_voltage)`` (/root/reference/src/mna/devices.jl:1003-1058).  A closure cannot cross a C ABI, so the
device table carries the expression as text -- ``"2*V(a) + V(b,c)**2"`` -- which is compiled here, once, to the
postfix program the stamp kernels interpret.  Grammar: numbers, ``V(node)`` / ``V(p, n)``, ``t`` / ``time``,
``+ - * / ** ^``, unary minus, ``min max pow exp log sqrt abs tanh sin cos``.  Constants only: a behavioural
source cannot reference a swept parameter (its program is shared by all instances of a batch); the per-instance
``scale`` parameter multiplies the result.
"""
import ast
import re

OP = {"const": 0, "v": 1, "time": 2, "add": 10, "sub": 11, "mul": 12, "div": 13, "pow": 14, "min": 15, "max": 16,
      "neg": 20, "exp": 21, "log": 22, "sqrt": 23, "abs": 24, "tanh": 25, "sin": 26, "cos": 27}
MAX_STACK = 16
_BIN = {ast.Add: "add", ast.Sub: "sub", ast.Mult: "mul", ast.Div: "div", ast.Pow: "pow", ast.BitXor: "pow"}
_FN1 = ("exp", "log", "sqrt", "abs", "tanh", "sin", "cos")
_FN2 = ("min", "max", "pow")


def _node_name(a):
    if isinstance(a, ast.Name):
        return a.id
    if isinstance(a, ast.Constant):
        return str(a.value)
    raise ValueError("V() takes node names")


def compile_expr(text):
    """-> list of tokens: ("const", x) | ("v", p_name, n_name) | ("time",) | (opname,)"""
    out = []

    def walk(e):
        if isinstance(e, ast.Constant) and isinstance(e.value, (int, float)):
            out.append(("const", float(e.value)))
        elif isinstance(e, ast.Name) and e.id.lower() in ("t", "time"):
            out.append(("time",))
        elif isinstance(e, ast.UnaryOp) and isinstance(e.op, ast.USub):
            walk(e.operand); out.append(("neg",))
        elif isinstance(e, ast.UnaryOp) and isinstance(e.op, ast.UAdd):
            walk(e.operand)
        elif isinstance(e, ast.BinOp) and type(e.op) in _BIN:
            walk(e.left); walk(e.right); out.append((_BIN[type(e.op)],))
        elif isinstance(e, ast.Call) and isinstance(e.func, ast.Name):
            f = e.func.id.lower()
            if f == "v" and len(e.args) in (1, 2):
                out.append(("v", _node_name(e.args[0]), _node_name(e.args[1]) if len(e.args) == 2 else "0"))
            elif f in _FN1 and len(e.args) == 1:
                walk(e.args[0]); out.append((f,))
            elif f in _FN2 and len(e.args) == 2:
                walk(e.args[0]); walk(e.args[1]); out.append((f,))
            else:
                raise ValueError("unsupported call %s/%d in behavioural expression" % (f, len(e.args)))
        else:
            raise ValueError("unsupported syntax in behavioural expression: %s" % ast.dump(e))

    # node names are arbitrary SPICE tokens ("in", "1", "net+"): quote them before handing the text to the parser
    quoted = re.sub(r"\b[vV]\(\s*([^(),\s]+)\s*(?:,\s*([^(),\s]+)\s*)?\)",
                    lambda m: "V(%r)" % m.group(1) if m.group(2) is None else "V(%r, %r)" % (m.group(1), m.group(2)), text.strip())
    walk(ast.parse(quoted, mode="eval").body)
    depth = peak = 0
    for tok in out:
        depth += 1 if tok[0] in ("const", "v", "time") else (-1 if OP[tok[0]] < OP["neg"] else 0)
        peak = max(peak, depth)
    if peak > MAX_STACK:
        raise ValueError("behavioural expression needs a stack deeper than %d" % MAX_STACK)
    return out


def encode(tokens, node_index):
    """Tokens -> doubles; ``node_index(name)`` gives the unknown index (-1 for ground)."""
    prog = []
    for tok in tokens:
        prog.append(float(OP[tok[0]]))
        if tok[0] == "const":
            prog.append(tok[1])
        elif tok[0] == "v":
            prog += [float(node_index(tok[1])), float(node_index(tok[2]))]
    return prog
