"""Verilog-A compact models on the GPU path: front end (frontend.py), host value evaluation (host_eval.py) and the HIP
stamp-function generator (hipgen.py).  The library is built with one generated function per module of ``MODEL_FILES``;
a model's id is its position in that list."""
import os

from .frontend import VAError, VAModule, parse_module
from . import host_eval

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
# the modules compiled into libcadnip_hip.so, in model-id order (csrc/build.sh passes the same list to hipgen.py)
MODEL_FILES = ("va_resistor.va", "va_capacitor.va", "va_diode.va", "va_sqmos.va", "va_dlim.va", "va_mos1l.va", "va_feat.va")

# Models whose Verilog-A sources are NOT part of this repository (third-party text inside the reference): the library carries
# their generated stamp functions (csrc/va_generated_ext.hpp, model ids after the built-in ones); the host side -- structure
# discovery, parameter defaults -- needs the source itself and looks for it in $CADNIP_VA_PATH (os.pathsep-separated directories)
# and then in the reference checkout.  (module name, file name, directories below the reference root)
_VAD, _PSP, _CMC = ("models/VADistillerModels.jl/va",), ("models/PSPModels.jl/va",), ("models/CMCModels.jl/va",)
EXTERNAL = (("PSP103VA", "psp103.va", _PSP), ("sp_bsim4v8", "bsim4v8.va", _VAD),
            # the rest of the reference's model packages (test/mna/vadistiller_integration.jl Tier 6 exercises each VADistiller model)
            ("sp_resistor", "resistor.va", _VAD), ("sp_capacitor", "capacitor.va", _VAD), ("sp_diode", "diode.va", _VAD),
            ("sp_bjt", "bjt.va", _VAD), ("sp_jfet1", "jfet1.va", _VAD), ("sp_jfet2", "jfet2.va", _VAD), ("sp_mes1", "mes1.va", _VAD),
            ("sp_mos1", "mos1.va", _VAD), ("sp_mos2", "mos2.va", _VAD), ("sp_mos3", "mos3.va", _VAD), ("sp_mos6", "mos6.va", _VAD),
            ("sp_mos9", "mos9.va", _VAD), ("sp_bsim3v3", "bsim3v3.va", _VAD), ("JUNCAP200", "juncap200.va", _PSP),
            ("bsimcmg", "bsimcmg.va", _CMC), ("sp_inductor", "inductor.va", _VAD), ("sp_vdmos", "vdmos.va", _VAD),
            ("NLVCR", "NLVCR.va", ("test",)),       # test/ddx.jl: ddx() with respect to a branch potential
            ("TMRoundTrip", "tm_1d.va", ("test/mna/fixtures/table_model",)), ("TM2D", "tm_2d.va", ("test/mna/fixtures/table_model",)))   # test/mna/table_model.jl
REFERENCE_ROOT = "/root/reference"

_cache = {}


def external_source(fn, subdirs):
    dirs = [d for d in os.environ.get("CADNIP_VA_PATH", "").split(os.pathsep) if d] + [os.path.join(REFERENCE_ROOT, d) for d in subdirs]
    for d in dirs:
        p = os.path.join(d, fn)
        if os.path.isfile(p):
            return p
    return None


def _builtin():
    if not _cache:
        for i, fn in enumerate(MODEL_FILES):
            m = parse_module(open(os.path.join(MODEL_DIR, fn)).read(), MODEL_DIR)
            _cache[m.name] = (i, m)
    return _cache


def _external(name):
    """(model id, VAModule) of an external model, parsed on first use (PSP103 / bsim4v8: seconds); None when its source is not at hand."""
    if name not in _ext_cache:
        from .frontend import parse_file
        j = next(k for k, e in enumerate(EXTERNAL) if e[0] == name)
        p = external_source(EXTERNAL[j][1], EXTERNAL[j][2])
        if p is None:
            _ext_cache[name] = None
        else:
            m = parse_file(p)
            if m.name != name:
                raise VAError("%s defines module %s, expected %s" % (p, m.name, name))
            _ext_cache[name] = (len(MODEL_FILES) + j, m)
    return _ext_cache[name]


_ext_cache = {}


def module_names():
    """lower-case name -> module name of everything a deck may instantiate: the built-in modules and the external models whose
    source is found (they are parsed only when used: ``get``)."""
    out = {nm.lower(): nm for nm in _builtin()}
    for name, fn, subdirs in EXTERNAL:
        if external_source(fn, subdirs) is not None:
            out[name.lower()] = name
    return out


def registry():
    """name -> (model id, VAModule): the built-in modules, then the external ones whose source is found (parses all of them)."""
    out = dict(_builtin())
    for name, _, _ in EXTERNAL:
        e = _external(name)
        if e is not None:
            out[name] = e
    return out


def get(name):
    try:
        if name in _builtin():
            return _builtin()[name]
        e = _external(name) if any(x[0] == name for x in EXTERNAL) else None
        if e is None:
            raise KeyError(name)
        return e
    except KeyError:
        ext = [e for e in EXTERNAL if e[0] == name]
        if ext:
            raise VAError("the library carries %s, but its Verilog-A source %s was not found (set CADNIP_VA_PATH to its directory): the host "
                          "side needs it for structure discovery and parameter defaults" % (name, ext[0][1])) from None
        raise VAError("no Verilog-A module %r is compiled into the library (have: %s)" % (name, ", ".join(module_names().values()))) from None
