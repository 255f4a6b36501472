"""Import shim: the product package lives in the directory ``cadnip.jl_amd/`` (the name the
build contract fixes); a dot is not importable, so this module loads that directory as the
package ``cadnip_jl_amd``.  ``import cadnip_jl_amd`` then behaves like a normal package import."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cadnip.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "cadnip_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cadnip_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
