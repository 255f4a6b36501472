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
EXTERNAL = (("PSP103VA", "psp103.va", ("models/PSPModels.jl/va",)),
            ("sp_bsim4v8", "bsim4v8.va", ("models/VADistillerModels.jl/va",)))
REFERENCE_ROOT = "/root/reference"

_cache = {}


def external_source(fn, subdirs):
    dirs = [d for d in os.environ.get("CADNIP_VA_PATH", "").split(os.pathsep) if d] + [os.path.join(REFERENCE_ROOT, d) for d in subdirs]
    for d in dirs:
        p = os.path.join(d, fn)
        if os.path.isfile(p):
            return p
    return None


def registry():
    """name -> (model id, VAModule): the built-in modules, then the external ones whose source is found."""
    if not _cache:
        from .frontend import parse_file
        for i, fn in enumerate(MODEL_FILES):
            m = parse_module(open(os.path.join(MODEL_DIR, fn)).read(), MODEL_DIR)
            _cache[m.name] = (i, m)
        for j, (name, fn, subdirs) in enumerate(EXTERNAL):
            p = external_source(fn, subdirs)
            if p is not None:
                m = parse_file(p)
                if m.name != name:
                    raise VAError("%s defines module %s, expected %s" % (p, m.name, name))
                _cache[m.name] = (len(MODEL_FILES) + j, m)
    return _cache


def get(name):
    try:
        return registry()[name]
    except KeyError:
        ext = [e for e in EXTERNAL if e[0] == name]
        if ext:
            raise VAError("the library carries %s, but its Verilog-A source %s was not found (set CADNIP_VA_PATH to its directory): the host "
                          "side needs it for structure discovery and parameter defaults" % (name, ext[0][1])) from None
        raise VAError("no Verilog-A module %r is compiled into the library (have: %s)" % (name, ", ".join(registry()))) from None
