"""Flattened device table -- the input of the GPU path.

The reference reaches the hot path through a generated Julia ``builder`` that calls
``stamp!`` once per instance in netlist order (/root/reference/src/spc/codegen.jl:3437-3518).
This build takes the same information as a plain device table: one row per instance with
its type, terminal names and parameters, in netlist order (V-sources first for SPICE decks,
codegen.jl:3130-3149).  The netlist front end itself is out of scope.

Parameter values may be numbers or ``Param("name")`` references that are resolved against
the circuit's sweepable parameters -- the role of ``.param`` / ``ParamLens`` in the reference
(/root/reference/src/sweeps.jl:417-424).
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple


@dataclass(frozen=True)
class Param:
    """Reference to a sweepable circuit parameter; ``scale`` and ``offset`` allow
    ``value = scale * params[name] + offset``."""
    name: str
    scale: float = 1.0
    offset: float = 0.0


@dataclass
class Device:
    type: str
    name: str
    nodes: Tuple[str, ...]
    params: Dict[str, Any] = field(default_factory=dict)
    wave: Optional[tuple] = None
    model: Optional[Dict[str, Any]] = None


class Circuit:
    """Ordered device table.  Methods mirror the SPICE element letters."""

    def __init__(self, title: str = ""):
        self.title = title
        self.devices: List[Device] = []

    def _add(self, ty, name, nodes, params=None, wave=None, model=None):
        self.devices.append(Device(ty, name, tuple(str(n) for n in nodes), dict(params or {}), wave, model))
        return self

    def R(self, name, p, n, r):
        return self._add("R", name, (p, n), {"r": r})

    def C(self, name, p, n, c):
        return self._add("C", name, (p, n), {"c": c})

    def L(self, name, p, n, l):
        return self._add("L", name, (p, n), {"l": l})

    def V(self, name, p, n, dc=0.0, wave=None, scale=1.0, ac=0.0):
        """``ac``: small-signal excitation (complex: magnitude and phase), stamped into b_ac only (devices.jl:659)."""
        return self._add("V", name, (p, n), {"dc": dc, "scale": scale, "ac": ac}, wave=wave)

    def I(self, name, p, n, dc=0.0, wave=None, scale=1.0, ac=0.0):
        return self._add("I", name, (p, n), {"dc": dc, "scale": scale, "ac": ac}, wave=wave)

    def E(self, name, op, on, ip, in_, gain):
        return self._add("E", name, (op, on, ip, in_), {"gain": gain})

    def G(self, name, op, on, ip, in_, gm):
        return self._add("G", name, (op, on, ip, in_), {"gm": gm})

    def H(self, name, op, on, ip, in_, rm):
        return self._add("H", name, (op, on, ip, in_), {"rm": rm})

    def F(self, name, op, on, ip, in_, gain):
        return self._add("F", name, (op, on, ip, in_), {"gain": gain})

    def BV(self, name, p, n, expr, scale=1.0):
        """BehavioralVoltageSource (devices.jl:1003-1029, stamp :1079-1102): V(p,n) = scale * expr, expr a string
        over V(node) / V(a,b) / t (bsource.py).  Stamped as a fixed source at the current iterate."""
        return self._add("BV", name, (p, n), {"expr": str(expr), "scale": scale})

    def BI(self, name, p, n, expr, scale=1.0):
        """BehavioralCurrentSource (devices.jl:1032-1058, stamp :1118-1131): scale * expr flows into p."""
        return self._add("BI", name, (p, n), {"expr": str(expr), "scale": scale})

    # KF / AF / FFE: flicker-noise coefficient and exponents of the reference's Diode, DiodeWithCap and SimpleMOSFET (devices.jl:1277-1279,
    # 1462-1464, 1625-1627); read by noise analysis only (no stamp depends on them)
    def D(self, name, p, n, Is=1e-14, Vt=0.026, n_=1.0, limit=True, KF=0.0, AF=1.0, FFE=1.0):
        return self._add("D", name, (p, n), {"Is": Is, "Vt": Vt, "n": n_, "limit": bool(limit), "KF": KF, "AF": AF, "FFE": FFE})

    def DCAP(self, name, p, n, Is=1e-14, Vt=0.026, n_=1.0, Cj0=1e-12, Vj=0.7, m=0.5, KF=0.0, AF=1.0, FFE=1.0):
        return self._add("DCAP", name, (p, n), {"Is": Is, "Vt": Vt, "n": n_, "Cj0": Cj0, "Vj": Vj, "m": m, "KF": KF, "AF": AF, "FFE": FFE})

    def SMOS(self, name, d, g, s, Vth=0.5, K=1e-3, lambda_=0.0, Cgd=1e-15, Cgs=1e-15, KF=0.0, AF=1.0, FFE=1.0):
        return self._add("SMOS", name, (d, g, s), {"Vth": Vth, "K": K, "lambda": lambda_, "Cgd": Cgd, "Cgs": Cgs, "KF": KF, "AF": AF, "FFE": FFE})

    def MOS1(self, name, d, g, s, b, model, m=1.0, **instance):
        """sp_mos1 (models/VADistillerModels.jl/va/mos1.va).  ``model`` holds the model-card
        parameters that were given; ``instance`` the given instance parameters (w, l, ad, ...)."""
        mp = dict(model)
        mp.update(instance)
        return self._add("MOS1", name, (d, g, s, b), {"m": m}, model=mp)

    def VA(self, name, module, nodes, m=1.0, **params):
        """Instance of a Verilog-A module that is compiled into the library (cadnip.jl_amd/va: ``va.registry()``);
        ``nodes`` in port order, ``params`` the given module parameters, ``m`` the multiplicity ($mfactor)."""
        mp = {k.lower(): v for k, v in params.items()}
        return self._add("VA:" + module, name, tuple(nodes), {"m": m}, model=mp)

    # neutral form used by tests to feed the oracle's builder (oracle/netlist_ref.py)
    def to_dicts(self, params=None):
        """Plain-dict form of the table.  With ``params`` every Param is evaluated to a number;
        without, plain Params are exported by name (affine ones cannot be)."""
        out = []
        for d in self.devices:
            e = {"type": d.type, "name": d.name, "nodes": list(d.nodes)}
            for k, v in d.params.items():
                e[k] = _plain(v, params)
            if d.wave is not None:
                e["wave"] = d.wave
            if d.model is not None:
                e["model"] = {k: _plain(v, params) for k, v in d.model.items()}
            out.append(e)
        return out


def _plain(v, params=None):
    if isinstance(v, Param):
        if params is not None:
            return float(resolve(v, params))
        if v.scale != 1.0 or v.offset != 0.0:
            raise ValueError("affine Param cannot be exported by name; pass params to evaluate it")
        return v.name
    return v


def resolve(v, params):
    """Evaluate a device parameter for one parameter set (numbers or numpy arrays)."""
    if isinstance(v, Param):
        return v.scale * params[v.name] + v.offset
    return v
