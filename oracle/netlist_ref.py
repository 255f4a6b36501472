"""TEST INFRASTRUCTURE (oracle) -- builder functions from a neutral device list.

The reference turns a netlist into a Julia ``builder(params, spec, t; x, ctx)`` that
calls ``stamp!`` once per instance in netlist order
(/root/reference/src/spc/codegen.jl:3437-3518).  ``make_builder`` does the same from
a list of plain dicts::

    {"type": "R", "name": "r1", "nodes": ["a", "b"], "r": 1e3}

Numeric fields may be a number or the name of a key of ``params`` (sweepable).
Wave fields: ``wave = ("pwl", ts, ys) | ("pulse", v1, v2, td, tr, tf, pw, per) |
("sin", vo, va, freq, td, theta, phase)``; ``scale`` multiplies the transient value.
"""
from . import devices_ref as D
from .mna_ref import MNAContext, ZERO_VECTOR
from .va_mos1_ref import Mos1Model, stamp_mos1
from .va_ref import stamp_va


def _num(v, params):
    if isinstance(v, str):
        return float(params[v])
    return v


class _Scaled:
    def __init__(self, wave, scale):
        self.wave = wave
        self.scale = scale

    def __call__(self, t):
        return self.scale * self.wave(t)

    def breakpoints(self):
        return self.wave.breakpoints()


def make_wave(spec, scale=1.0):
    if spec is None:
        return None
    kind = spec[0]
    if kind == "pwl":
        w = D.PWLWave(spec[1], spec[2])
    elif kind == "pulse":
        w = D.PulseWave(*spec[1:])
    elif kind == "sin":
        w = D.SinWave(*spec[1:])
    else:
        raise ValueError(kind)
    return w if scale == 1.0 else _Scaled(w, scale)


def behavioral_fn(expr, t, scale=1.0):
    """``value_fn(get_voltage)`` of a behavioural source (devices.jl:1003-1058) from expression text: evaluated by the
    Python interpreter itself, node names quoted first (the product compiles the same text to a postfix program)."""
    import math
    import re
    src = re.sub(r"\b[vV]\(([^()]*)\)", lambda m: "V(%s)" % ", ".join(repr(a.strip()) for a in m.group(1).split(",")), expr)
    src = src.replace("^", "**")
    code = compile(src, "<bsource>", "eval")

    def value_fn(get_voltage):
        ns = {"V": lambda a, b="0": get_voltage(a) - get_voltage(b), "t": t, "time": t, "exp": math.exp, "log": math.log,
              "sqrt": math.sqrt, "abs": abs, "tanh": math.tanh, "sin": math.sin, "cos": math.cos, "min": min, "max": max,
              "pow": math.pow, "__builtins__": {}}
        return scale * eval(code, ns)
    return value_fn


def make_builder(devices):
    def builder(params, spec, t, x=ZERO_VECTOR, ctx=None):
        if ctx is None:
            ctx = MNAContext()
        for dev in devices:
            ty = dev["type"]
            nodes = [ctx.get_node(nm) for nm in dev["nodes"]]
            g = lambda k, default=None: _num(dev.get(k, default), params)
            name = dev.get("name", ty)
            if ty == "R":
                D.stamp_resistor(ctx, nodes[0], nodes[1], g("r"), name)
            elif ty == "C":
                D.stamp_capacitor(ctx, nodes[0], nodes[1], g("c"))
            elif ty == "L":
                D.stamp_inductor(ctx, nodes[0], nodes[1], g("l"), name)
            elif ty == "V":
                w = make_wave(dev.get("wave"), g("scale", 1.0))
                D.stamp_vsource(ctx, nodes[0], nodes[1], g("dc", 0.0), w, t, spec.mode, name)
            elif ty == "I":
                w = make_wave(dev.get("wave"), g("scale", 1.0))
                D.stamp_isource(ctx, nodes[0], nodes[1], g("dc", 0.0), w, t, spec.mode, name)
            elif ty == "E":
                D.stamp_vcvs(ctx, nodes[0], nodes[1], nodes[2], nodes[3], g("gain"), name)
            elif ty == "G":
                D.stamp_vccs(ctx, nodes[0], nodes[1], nodes[2], nodes[3], g("gm"))
            elif ty == "H":
                D.stamp_ccvs(ctx, nodes[0], nodes[1], nodes[2], nodes[3], g("rm"), name)
            elif ty == "F":
                D.stamp_cccs(ctx, nodes[0], nodes[1], nodes[2], nodes[3], g("gain"), name)
            elif ty in ("BV", "BI"):
                def get_voltage(nm, ctx=ctx, x=x):
                    i = ctx.get_node(nm)
                    return 0.0 if i == 0 else float(x[i - 1])
                fn = behavioral_fn(dev["expr"], t, g("scale", 1.0))
                if ty == "BV":
                    D.stamp_behavioral_vsource(ctx, nodes[0], nodes[1], fn, name, get_voltage)
                else:
                    D.stamp_behavioral_isource(ctx, nodes[0], nodes[1], fn, get_voltage)
            elif ty == "D":
                D.stamp_diode(ctx, nodes[0], nodes[1], x, g("Is", 1e-14), g("Vt", 0.026), g("n", 1.0),
                              bool(dev.get("limit", True)), name, g("KF", 0.0), g("AF", 1.0), g("FFE", 1.0))
            elif ty == "DCAP":
                D.stamp_diode_with_cap(ctx, nodes[0], nodes[1], x, g("Is", 1e-14), g("Vt", 0.026), g("n", 1.0),
                                       g("Cj0", 1e-12), g("Vj", 0.7), g("m", 0.5), name, g("KF", 0.0), g("AF", 1.0), g("FFE", 1.0))
            elif ty == "SMOS":
                D.stamp_simple_mosfet(ctx, nodes[0], nodes[1], nodes[2], x, g("Vth", 0.5), g("K", 1e-3),
                                      g("lambda", 0.0), g("Cgd", 1e-15), g("Cgs", 1e-15), name, g("KF", 0.0), g("AF", 1.0), g("FFE", 1.0))
            elif ty == "MOS1":
                mp = {k: _num(v, params) for k, v in dev["model"].items()}
                stamp_mos1(ctx, Mos1Model(**mp), nodes[0], nodes[1], nodes[2], nodes[3], x, spec, name,
                           mfactor=g("m", 1.0))
            elif ty.startswith("VA:"):
                # a generated Verilog-A module: the syntax tree and the parameter defaults come from the product's front end
                from cadnip_jl_amd import va
                mod = va.get(ty[3:])[1]
                par = va.host_eval.defaults(mod, {k: _num(v, params) for k, v in dev["model"].items()})
                stamp_va(ctx, mod, nodes, x, par, spec, name, mfactor=g("m", 1.0), given=set(dev["model"]))
            else:
                raise ValueError("unknown device type %r" % ty)
        return ctx
    return builder
