"""Verilog-A compact models on the GPU path: front end (frontend.py), host value evaluation (host_eval.py) and the HIP
stamp-function generator (hipgen.py).  The library is built with one generated function per module of ``MODEL_FILES``;
a model's id is its position in that list."""
import os

from .frontend import VAError, VAModule, parse_module
from . import host_eval

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
# the modules compiled into libcadnip_hip.so, in model-id order (csrc/build.sh passes the same list to hipgen.py)
MODEL_FILES = ("va_resistor.va", "va_capacitor.va", "va_diode.va", "va_sqmos.va", "va_dlim.va", "va_mos1l.va", "va_feat.va")

_cache = {}


def registry():
    """name -> (model id, VAModule) of the built-in modules."""
    if not _cache:
        for i, fn in enumerate(MODEL_FILES):
            m = parse_module(open(os.path.join(MODEL_DIR, fn)).read(), MODEL_DIR)
            _cache[m.name] = (i, m)
    return _cache


def get(name):
    try:
        return registry()[name]
    except KeyError:
        raise VAError("no Verilog-A module %r is compiled into the library (have: %s)" % (name, ", ".join(registry()))) from None
