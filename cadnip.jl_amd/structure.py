"""Structure discovery for the GPU path -- the host-side, run-once half of the hot path.

Reproduces what the reference obtains from running the generated builder on an
``MNAContext`` and compiling it (/root/reference/src/mna/context.jl,
/root/reference/src/mna/precompile.jl:312-443), directly from the device table:

* unknown layout ``[V | I | q | v_lim]`` (context.jl:436-438, typed-index resolution
  :577-581), nodes numbered by first use, currents / charges / limits by allocation order;
* the COO stamp streams in the reference's order (instance order, then the fixed local
  stamp order of each ``stamp!``), ground rows/cols skipped before the position counter
  (value_only.jl:395-397);
* the unified G u C pattern (precompile.jl:413-421) -- stored CSR here, with the map to
  the reference's CSC ``nzval`` order for read-back;
* slot -> nz gather lists that replace the positional COO->nz maps
  (precompile.jl:253-283): every local stamp of every device owns one slot of a
  per-instance slot buffer, and each nz entry sums its slots in COO order.

The result is exactly what ``CadnipStructure`` (include/cadnip_hip.h) carries.
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from . import bsource
from . import mos1_params as m1
from . import va
from .circuit import Circuit, resolve

# device type ids: keep in sync with CadnipDeviceType in include/cadnip_hip.h
TYPE_ID = {"R": 0, "C": 1, "L": 2, "V": 3, "I": 4, "E": 5, "G": 6, "H": 7, "F": 8,
           "D": 9, "DCAP": 10, "SMOS": 11, "MOS1": 12, "BV": 13, "BI": 14}
TYPE_ID_VA = 15          # CADNIP_DEV_VA: every "VA:<module>" block (the module is ipar row 0)
WAVE_DC, WAVE_PWL, WAVE_PULSE, WAVE_SIN = 0, 1, 2, 3


def type_id(ty):
    return TYPE_ID_VA if ty.startswith("VA:") else TYPE_ID[ty]


def shape_of(ty):
    """(n_local_nodes, n_g, n_c, n_b, n_par, n_ipar) of a device type; generated Verilog-A modules bring their own."""
    if ty.startswith("VA:"):
        return va.get(ty[3:])[1].shape()
    return SHAPE[ty]

# (n_local_nodes, n_g, n_c, n_b, n_par, n_ipar) per type
SHAPE = {
    "R": (2, 4, 0, 0, 1, 0), "C": (2, 0, 4, 0, 1, 0), "L": (3, 4, 1, 0, 1, 0),
    "V": (3, 4, 0, 1, 2, 3), "I": (2, 0, 0, 2, 2, 3), "E": (5, 6, 0, 0, 1, 0),
    "G": (4, 4, 0, 0, 1, 0), "H": (6, 9, 0, 0, 1, 0), "F": (5, 6, 0, 0, 1, 0),
    "D": (3, 7, 0, 2, 3, 1), "DCAP": (2, 4, 4, 2, 5, 0), "SMOS": (3, 6, 8, 2, 5, 0),
    "MOS1": (14, 76, 28, 10, m1.NPAR, 1),
    "BV": (3, 4, 0, 1, 1, 2), "BI": (2, 0, 0, 2, 1, 2),
}

GND = -1


@dataclass
class Block:
    type: str
    count: int
    dev_index: List[int]             # positions in circuit.devices
    nodes: np.ndarray                # [n_local, count] int32 unknown index or -1
    ipar: np.ndarray                 # [n_ipar, count] int32
    g_base: int = 0
    c_base: int = 0
    b_base: int = 0
    n_g: int = 0
    n_c: int = 0
    n_b: int = 0
    n_par: int = 0


@dataclass
class Structure:
    n: int
    n_nodes: int
    n_currents: int
    n_charges: int
    n_limits: int
    node_names: List[str]
    current_names: List[str]
    charge_names: List[str]
    limit_names: List[str]
    rowptr: np.ndarray
    colidx: np.ndarray
    to_ref_nz: np.ndarray
    ref_colptr: np.ndarray
    ref_rowval: np.ndarray
    blocks: List[Block]
    wave_data: np.ndarray
    ns_g: int
    ns_c: int
    ns_b: int
    g_ptr: np.ndarray
    g_slots: np.ndarray
    c_ptr: np.ndarray
    c_slots: np.ndarray
    b_ptr: np.ndarray
    b_slots: np.ndarray
    diag_nz: np.ndarray
    limit_init: np.ndarray
    breakpoints: list = field(default_factory=list)
    n_coo_g: int = 0
    n_coo_c: int = 0
    n_coo_b: int = 0
    opinfo: list = field(default_factory=list)     # per device: terminals, local rows per terminal, local stamp program (opinfo.py)
    mos1_vdep: tuple = (False, False, False, False)

    @property
    def nnz(self):
        return int(self.colidx.shape[0])

    def signature(self):
        """hash of everything a device handle is built from except parameter values: two sweep points with equal signatures
        can share one resident batch"""
        import hashlib
        h = hashlib.sha1()
        h.update(repr((self.n, self.n_nodes, self.n_currents, self.n_charges, self.n_limits, self.ns_g, self.ns_c, self.ns_b,
                       tuple(self.node_names), tuple(self.current_names), tuple(self.charge_names), tuple(self.limit_names))).encode())
        for a in (self.rowptr, self.colidx, self.g_ptr, self.g_slots, self.c_ptr, self.c_slots, self.b_ptr, self.b_slots):
            h.update(np.ascontiguousarray(a, dtype=np.int64).tobytes())
        for b in self.blocks:
            h.update(repr((b.type, b.count, b.n_par, b.n_g, b.n_c, b.n_b)).encode())
            h.update(np.ascontiguousarray(b.nodes, dtype=np.int64).tobytes())
            h.update(np.ascontiguousarray(b.ipar, dtype=np.int64).tobytes())
        return h.hexdigest()

    def index_of(self, name):
        """sol[:name] lookup order nodes -> currents -> charges -> limits (build.jl:421-457)."""
        if name in self.node_names:
            return self.node_names.index(name)
        if name in self.current_names:
            return self.n_nodes + self.current_names.index(name)
        if name in self.charge_names:
            return self.n_nodes + self.n_currents + self.charge_names.index(name)
        if name in self.limit_names:
            return self.n_nodes + self.n_currents + self.n_charges + self.limit_names.index(name)
        raise KeyError(name)

    def differential_mask(self):
        """1.0 for unknowns that appear differentiated (columns of C with at least one stamp), 0.0 for
        algebraic ones.  Column-wise counterpart of detect_differential_vars (solve.jl:2041-2058)."""
        m = np.zeros(self.n)
        has = np.diff(self.c_ptr) > 0
        m[np.unique(self.colidx[has])] = 1.0
        return m

    def state_abstol(self, vntol=1e-6, iabstol=1e-12, chgtol=1e-14):
        """build.jl:276-283."""
        tol = np.empty(self.n)
        a = self.n_nodes
        b = a + self.n_currents
        c = b + self.n_charges
        tol[:a] = vntol
        tol[a:b] = iabstol
        tol[b:c] = chgtol
        tol[c:] = vntol
        return tol


_ST_ARRAYS = ("rowptr", "colidx", "to_ref_nz", "ref_colptr", "ref_rowval", "wave_data", "g_ptr", "g_slots", "c_ptr", "c_slots", "b_ptr", "b_slots",
              "diag_nz", "limit_init")
_ST_SCALARS = ("n", "n_nodes", "n_currents", "n_charges", "n_limits", "ns_g", "ns_c", "ns_b", "n_coo_g", "n_coo_c", "n_coo_b")


def save_structure(st: Structure, path, **extra):
    """Structure -> one .npz (arrays as they are, names / blocks / operating-point table as JSON); ``extra``: further arrays to
    keep beside it (packed parameters, fixture states).  A structure discovered where a model's Verilog-A source is at hand can
    then drive the GPU path where it is not (the library carries the generated code, not the source)."""
    import json
    meta = {k: int(getattr(st, k)) for k in _ST_SCALARS}
    meta.update(node_names=st.node_names, current_names=st.current_names, charge_names=st.charge_names, limit_names=st.limit_names,
                breakpoints=st.breakpoints, opinfo=st.opinfo, mos1_vdep=list(st.mos1_vdep),
                blocks=[dict(type=b.type, count=b.count, dev_index=list(map(int, b.dev_index)), g_base=b.g_base, c_base=b.c_base, b_base=b.b_base,
                             n_g=b.n_g, n_c=b.n_c, n_b=b.n_b, n_par=b.n_par) for b in st.blocks])
    arrays = {k: np.asarray(getattr(st, k)) for k in _ST_ARRAYS}
    for i, b in enumerate(st.blocks):
        arrays["blk%d_nodes" % i] = b.nodes
        arrays["blk%d_ipar" % i] = b.ipar
    arrays.update({"x_" + k: np.asarray(v) for k, v in extra.items()})
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta, default=lambda o: o.tolist() if hasattr(o, "tolist") else list(o)).encode(), dtype=np.uint8), **arrays)


def load_structure(path):
    """-> (Structure, extra arrays) as written by save_structure."""
    import json
    z = np.load(path, allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    blocks = [Block(b["type"], b["count"], b["dev_index"], z["blk%d_nodes" % i], z["blk%d_ipar" % i], b["g_base"], b["c_base"], b["b_base"],
                    b["n_g"], b["n_c"], b["n_b"], b["n_par"]) for i, b in enumerate(meta["blocks"])]

    def tup(o):     # JSON turns the tuples of the operating-point programs into lists
        return tuple(tup(x) for x in o) if isinstance(o, list) else o
    opinfo = [dict(d, prog=[tup(r) for r in d["prog"]], terminals=tuple(d["terminals"]), nodes=[tup(x) for x in d["nodes"]]) for d in meta["opinfo"]]
    st = Structure(blocks=blocks, node_names=meta["node_names"], current_names=meta["current_names"], charge_names=meta["charge_names"],
                   limit_names=meta["limit_names"], breakpoints=[tuple(b) for b in meta["breakpoints"]], opinfo=opinfo, mos1_vdep=tuple(meta["mos1_vdep"]),
                   **{k: meta[k] for k in _ST_SCALARS}, **{k: z[k] for k in _ST_ARRAYS})
    return st, {k[2:]: z[k] for k in z.files if k.startswith("x_")}


# local stamp programs: (stream, local_slot, row_local, col_local); order = the reference's
# stamp order inside each stamp! method (devices.jl line numbers in include/cadnip_hip.h)
def _prog_conductance(p, n, k0=0):
    return [("G", k0, p, p), ("G", k0 + 1, p, n), ("G", k0 + 2, n, p), ("G", k0 + 3, n, n)]


def _prog_cap(p, n, k0=0):
    return [("C", k0, p, p), ("C", k0 + 1, p, n), ("C", k0 + 2, n, p), ("C", k0 + 3, n, n)]


def _branch_pairs(p, n, I):
    return [("G", 0, p, I), ("G", 1, n, I), ("G", 2, I, p), ("G", 3, I, n)]


PROGRAMS = {
    "R": _prog_conductance(0, 1),
    "C": _prog_cap(0, 1),
    "L": _branch_pairs(0, 1, 2) + [("C", 0, 2, 2)],
    "V": _branch_pairs(0, 1, 2) + [("b", 0, 2, None)],
    "I": [("b", 0, 0, None), ("b", 1, 1, None)],
    "E": [("G", 0, 0, 4), ("G", 1, 1, 4), ("G", 2, 4, 0), ("G", 3, 4, 1), ("G", 4, 4, 2), ("G", 5, 4, 3)],
    "G": [("G", 0, 0, 2), ("G", 1, 0, 3), ("G", 2, 1, 2), ("G", 3, 1, 3)],
    "H": [("G", 0, 2, 4), ("G", 1, 3, 4), ("G", 2, 4, 2), ("G", 3, 4, 3), ("G", 4, 0, 5), ("G", 5, 1, 5),
          ("G", 6, 5, 0), ("G", 7, 5, 1), ("G", 8, 5, 4)],
    "F": [("G", 0, 2, 4), ("G", 1, 3, 4), ("G", 2, 4, 2), ("G", 3, 4, 3), ("G", 4, 0, 4), ("G", 5, 1, 4)],
    "D": [("G", 0, 2, 2), ("G", 1, 2, 0), ("G", 2, 2, 1)] + _prog_conductance(0, 1, 3) + [("b", 0, 0, None), ("b", 1, 1, None)],
    "DCAP": _prog_conductance(0, 1) + [("b", 0, 0, None), ("b", 1, 1, None)] + _prog_cap(0, 1),
    "BV": _branch_pairs(0, 1, 2) + [("b", 0, 2, None)],       # devices.jl:1079-1102
    "BI": [("b", 0, 0, None), ("b", 1, 1, None)],             # devices.jl:1118-1131
    "SMOS": [("G", 0, 0, 0), ("G", 1, 0, 1), ("G", 2, 0, 2), ("G", 3, 2, 0), ("G", 4, 2, 1), ("G", 5, 2, 2),
             ("b", 0, 0, None), ("b", 1, 2, None)] + _prog_cap(1, 2, 0) + _prog_cap(1, 0, 4),
}

# sp_mos1 local node order: d g s b d_int s_int | lim(g,s_int) lim(d_int,s_int) lim(b,s_int) lim(b,d_int) | q_g q_b q_dint q_sint
M1_BRANCH_NODE = (0, 1, 2, 3, 4, 5)     # I(d) I(g) I(s) I(b) I(d_int) I(s_int)   (mos1.va:1164-1169)
M1_REACTIVE = (1, 3, 4, 5)              # branches whose contribution carries ddt()
M1_LIMIT_PN = ((1, 5), (4, 5), (3, 5), (3, 4))


def mos1_program(vdep):
    """Stamp order of the generated sp_mos1 stamp! (vasim.jl:3110-3138, 3319-3521).  ``vdep[r]``
    says whether reactive branch r uses the charge-state formulation (vasim.jl:3433-3472)."""
    prog = []
    for lb, (p, n) in enumerate(M1_LIMIT_PN):
        l = 6 + lb
        prog += [("G", 3 * lb, l, l), ("G", 3 * lb + 1, l, p), ("G", 3 * lb + 2, l, n)]
    for br in range(6):
        p = M1_BRANCH_NODE[br]
        for k in range(6):
            prog.append(("G", 12 + 6 * br + k, p, k))
        if br in M1_REACTIVE:
            r = M1_REACTIVE.index(br)
            if vdep[r]:
                q = 10 + r
                prog.append(("C", r, p, q))
                prog.append(("G", 48 + 7 * r, q, q))
                for k in range(6):
                    prog.append(("G", 48 + 7 * r + 1 + k, q, k))
                prog.append(("b", 6 + r, q, None))
            else:
                for k in range(6):
                    prog.append(("C", 4 + 6 * r + k, p, k))
        prog.append(("b", br, p, None))
    return prog


class _Alloc:
    """Typed-index allocation with late resolution (context.jl:47-110, 577-581)."""

    def __init__(self):
        self.node_names, self.node_idx = [], {}
        self.current_names, self.charge_names, self.limit_names = [], [], []
        self.limit_init = []

    def node(self, name):
        if name in ("0", "gnd", "gnd!"):
            return GND
        i = self.node_idx.get(name)
        if i is None:
            i = len(self.node_names)
            self.node_idx[name] = i
            self.node_names.append(name)
        return ("n", i)

    def current(self, name):
        self.current_names.append(name)
        return ("c", len(self.current_names) - 1)

    def charge(self, name):
        self.charge_names.append(name)
        return ("q", len(self.charge_names) - 1)

    def limit(self, name, init):
        self.limit_names.append(name)
        self.limit_init.append(float(init))
        return ("l", len(self.limit_names) - 1)

    def resolve(self, t):
        if t == GND:
            return -1
        kind, k = t
        nn, nc, nq = len(self.node_names), len(self.current_names), len(self.charge_names)
        return {"n": k, "c": nn + k, "q": nn + nc + k, "l": nn + nc + nq + k}[kind]


def _wave_ipar(wave, wave_data):
    if wave is None:
        return (WAVE_DC, 0, 0)
    kind = wave[0]
    off = len(wave_data)
    if kind == "pwl":
        ts, ys = list(map(float, wave[1])), list(map(float, wave[2]))
        assert len(ts) == len(ys)
        wave_data.extend(ts)
        wave_data.extend(ys)
        return (WAVE_PWL, off, len(ts))
    if kind == "pulse":
        wave_data.extend(map(float, wave[1:8]))
        return (WAVE_PULSE, off, 7)
    if kind == "sin":
        vals = list(map(float, wave[1:])) + [0.0] * (6 - len(wave[1:]))
        wave_data.extend(vals[:6])
        return (WAVE_SIN, off, 6)
    raise ValueError("unknown wave kind %r" % (kind,))


def wave_breakpoints(wave):
    """devices.jl:145, 180, 211-214."""
    if wave is None:
        return None
    if wave[0] == "pwl":
        return ("list", [float(t) for t in wave[1]])
    if wave[0] == "pulse":
        v1, v2, td, tr, tf, pw, per = map(float, wave[1:8])
        edges = [td, td + tr, td + tr + pw, td + tr + pw + tf]
        return ("periodic", edges, per) if per > 0 else ("list", edges)
    if wave[0] == "sin":
        td = float(wave[4]) if len(wave) > 4 else 0.0
        return ("list", [td]) if td > 0 else None
    return None


def expand_breakpoints(specs, tspan, max_points=100000):
    """solve.jl:1847-1935: sorted, ULP-deduplicated stop times strictly inside tspan."""
    import math
    t0, t1 = float(tspan[0]), float(tspan[1])
    out = []
    for s in specs:
        if s is None or not s[1]:
            continue
        if s[0] == "list":
            out += [t for t in s[1] if t0 < t < t1]
        else:
            _, times, period = s
            k0 = int(min(max(math.floor((t0 - max(times)) / period), 0.0), 1e15))
            k1 = int(min(max(math.ceil((t1 - min(times)) / period), -1.0), 1e15))
            k1 = min(k1, k0 + max_points - 1)
            for k in range(k0, k1 + 1):
                out += [t + k * period for t in times if t0 < t + k * period < t1]
    if not out:
        return out
    out.sort()
    del out[max_points:]
    ded = [out[0]]
    for t in out[1:]:
        if t - ded[-1] > 4 * max(math.ulp(ded[-1]), math.ulp(t)):
            ded.append(t)
    return ded


def detect_mos1_vdep(circuit: Circuit, params: Dict[str, float], seed: int = 0xDEADBEEF):
    """Voltage-dependent-charge detection for the sp_mos1 instances, as the reference does it:
    five builder passes (the first at x = 0, the rest at random x in [-1, 1]) comparing the
    apparent capacitance Q/V of each reactive branch between consecutive evaluations
    (build_with_detection, solve.jl:1793-1822; detect_or_cached!, contrib.jl:214-257).  Every VA
    stamp! resets the detection counter (vasim.jl:3926), so the cache holds one entry per
    reactive-branch *position* and all instances share it -- including the side effects:
    consecutive evaluations belong to different devices, and a branch whose probe node is
    ground (V = 0) never takes part in a comparison.  The result is one flag per position.
    The probe points come from numpy's generator instead of Julia's MersenneTwister."""
    mos = [d for d in circuit.devices if d.type == "MOS1"]
    if not mos:
        return (False, False, False, False)
    rng = np.random.default_rng(seed)
    names = {}
    for d in circuit.devices:
        for nm in d.nodes:
            if nm not in ("0", "gnd", "gnd!") and nm not in names:
                names[nm] = len(names)
    derived = []
    for d in mos:
        given = {k: resolve(v, params) for k, v in d.model.items()}
        der, _ = m1.derive(given, 27.0, 27.0, 1e-12, mfactor=float(np.asarray(resolve(d.params["m"], params)).flat[0]))
        derived.append((der[:, 0], m1.short_circuits(given)))
    is_vdep, Qs, Vs = [], [], []
    for p in range(5):
        xv = np.zeros(len(names)) if p == 0 else (rng.random(len(names)) - 0.5) * 2.0
        for d, (P, (sc_d, sc_s)) in zip(mos, derived):
            v = [0.0 if nm in ("0", "gnd", "gnd!") else xv[names[nm]] for nm in d.nodes]
            # internal nodes that are not collapsed are extra unknowns: random like everything else
            vdi = v[0] if sc_d else (0.0 if p == 0 else (rng.random() - 0.5) * 2.0)
            vsi = v[2] if sc_s else (0.0 if p == 0 else (rng.random() - 0.5) * 2.0)
            vold = [0.0] * 4 if p == 0 else list((rng.random(4) - 0.5) * 2.0)
            q = m1.host_charges(P, (v[0], v[1], v[2], v[3], vdi, vsi), vold)
            vbranch = (v[1], v[3], vdi, vsi)
            for pos in range(4):     # detection counter restarts per device (vasim.jl:3926)
                V, Q = float(vbranch[pos]), float(q[pos])
                if pos >= len(Qs):
                    is_vdep.append(False); Qs.append(Q); Vs.append(V)
                    continue
                if abs(V) > 1e-6 and abs(Vs[pos]) > 1e-6:
                    Cc, Cs = Q / V, Qs[pos] / Vs[pos]
                    diff, mx = abs(Cc - Cs), max(abs(Cc), abs(Cs))
                    if diff > 1e-15 and (mx < 1e-30 or diff / mx > 1e-6):
                        is_vdep[pos] = True
                Qs[pos], Vs[pos] = Q, V
    return tuple(is_vdep)


_inst_cache: Dict[tuple, tuple] = {}


def va_instance(mod, dev, params):
    """(parameter values, alias map, shorts_on, active branches) of one Verilog-A instance for parameter set ``params``
    (va/host_eval.py: instance_structure); cached per (module, given parameter values): a deck instantiates few distinct cards."""
    given = {k: float(np.asarray(resolve(v, params)).flat[0]) for k, v in dev.model.items()}
    key = (mod.name, id(mod), tuple(sorted(given.items())))
    hit = _inst_cache.get(key)
    if hit is None:
        par = va.host_eval.defaults(mod, given)
        hit = (par,) + tuple(mod.instance_structure(par, set(dev.model)))
        if len(_inst_cache) > 4096:
            _inst_cache.clear()
        _inst_cache[key] = hit
    return hit


def _shorts_connected(mod, shorts_on, nodes):
    """``shorts_on`` of the instance's card, without the two-node potential contributions whose nets the CIRCUIT has already made one
    unknown (both tied to ground, or to one net): the reference skips them (vasim.jl:2364, 3765 `if p_node != n_node`).  ``nodes``: the
    instance's global unknowns, ports then internal nodes."""
    out = list(shorts_on)
    for j, si in enumerate(mod.vshorts):
        if out[j] and mod.short_kind[si] != "named":
            a, b = mod.shorts[si][0], mod.shorts[si][1]
            if nodes[a] == (nodes[b] if b >= 0 else GND):
                if mod.short_kind[si] == "top":
                    raise ValueError("%s: V(%s,%s) <+ ... at the top level of the analog block joins two nets the circuit has already made one" % (
                        mod.name, mod.nodes[a], mod.nodes[b] if b >= 0 else "0"))
                out[j] = False
    return out


def detect_va_vdep(circuit: Circuit, params: Dict[str, float], seed: int = 0xDEADBEEF):
    """Per Verilog-A module in the circuit: which reactive branches use a charge unknown.  Emulates the reference's
    detection run (build_with_detection, solve.jl:1793-1822) pass by pass: five builder passes, the first at x = 0, pass
    k > 0 at ``x = 2 (rand(known_size) - 0.5)`` where ``known_size`` is the system size after the previous pass (charge
    unknowns allocated so far included); every VA stamp! restarts the detection position counter (vasim.jl:3926), so all
    instances -- of all modules -- share one cache indexed by reactive-branch *position*, and a position compares
    consecutive evaluations whichever device they belong to (detect_or_cached!, contrib.jl:214-257).  A flag, once set,
    stays.  The structure is the one of the last pass.  Probe points come from numpy's generator (the reference uses
    Julia's MersenneTwister); sp_mos1 instances in the same circuit keep their own detection (detect_mos1_vdep)."""
    devs = [d for d in circuit.devices if d.type.startswith("VA:")]
    if not devs:
        return {}
    # unknown numbering of the builder: nodes by first use (external nets in instance order, internal nodes as their
    # instance is stamped); the counts of the other unknown kinds fix where x ends
    A = _Alloc()
    n_cur = n_lim = n_m1q = 0
    m1v = detect_mos1_vdep(circuit, params)
    internal = {}
    for d in circuit.devices:
        for nm in d.nodes:
            A.node(nm)
        ty = d.type
        if ty in ("L", "V", "BV", "E", "F"):
            n_cur += 1
        elif ty == "H":
            n_cur += 2
        elif ty in ("BV", "BI"):
            for tok in bsource.compile_expr(d.params["expr"]):
                if tok[0] == "v":
                    A.node(tok[1]); A.node(tok[2])
        elif ty == "D" and bool(d.params.get("limit", True)):
            n_lim += 1
        elif ty == "MOS1":
            given = {k: resolve(v, params) for k, v in d.model.items()}
            sc_d, sc_s = m1.short_circuits(given)
            if not sc_d:
                A.node("%s_sp_mos1_d_int" % d.name)
            if not sc_s:
                A.node("%s_sp_mos1_s_int" % d.name)
            n_lim += 4
            n_m1q += sum(m1v)
        elif ty.startswith("VA:"):
            mod = va.get(ty[3:])[1]
            _, alias, shorts_on, _ = va_instance(mod, d, params)
            ext = [A.node(nm) for nm in d.nodes]
            loc = list(ext) + [None] * mod.n_internal
            for k in range(len(mod.ports), mod.n_nodes):
                if k not in alias:
                    loc[k] = A.node("%s_%s_%s" % (d.name, mod.name, mod.nodes[k]))
            for k in range(len(mod.ports), mod.n_nodes):
                if k in alias:
                    loc[k] = loc[alias[k]] if alias[k] >= 0 else A.node("0")       # V(a) <+ 0: the internal node is ground
            internal[d.name] = loc[len(mod.ports):]
            n_cur += sum(_shorts_connected(mod, shorts_on, loc))
            n_lim += len(mod.limit_branches)
    n_nodes = len(A.node_names)
    rng = np.random.default_rng(seed)
    is_vdep, Qs, Vs = [], [], []
    n_q_prev = 0
    last = {}
    for p in range(5):
        x = np.zeros(0) if p == 0 else (rng.random(n_nodes + n_cur + n_m1q + n_q_prev + n_lim) - 0.5) * 2.0

        def xat(i):      # 0-based read, tolerant of a short x (vasim.jl:3123-3133)
            return float(x[i]) if 0 <= i < len(x) else 0.0
        # the builder allocates as it goes: an index resolved in the middle of a pass uses the counts reached so far
        # (context.jl:577-581), which is where a $limit preamble finds its `vold`
        seen_nodes, cur_sf, q_sf, lim_sf = set(), 0, 0, 0
        for d in circuit.devices:
            ty = d.type
            for nm in d.nodes:
                if nm not in ("0", "gnd", "gnd!"):
                    seen_nodes.add(nm)
            if ty in ("L", "V", "BV", "E", "F"):
                cur_sf += 1
            elif ty == "H":
                cur_sf += 2
            elif ty == "D" and bool(d.params.get("limit", True)):
                lim_sf += 1
            elif ty == "MOS1":
                given = {k: resolve(v, params) for k, v in d.model.items()}
                sc_d, sc_s = m1.short_circuits(given)
                seen_nodes.update(nm for nm, sc in (("%s_sp_mos1_d_int" % d.name, sc_d), ("%s_sp_mos1_s_int" % d.name, sc_s)) if not sc)
                lim_sf += 4
                q_sf += sum(m1v)
            if ty in ("BV", "BI"):
                for tok in bsource.compile_expr(d.params["expr"]):
                    if tok[0] == "v":
                        seen_nodes.update(nm for nm in (tok[1], tok[2]) if nm not in ("0", "gnd", "gnd!"))
            if not ty.startswith("VA:"):
                continue
            mod = va.get(ty[3:])[1]
            par, _, shorts_on, active = va_instance(mod, d, params)
            idx = [A.node(nm) for nm in d.nodes] + internal[d.name]
            seen_nodes.update(A.node_names[t[1]] for t in internal[d.name] if t != GND)
            V = [0.0 if t == GND else xat(t[1]) for t in idx]
            vold = []
            for l in range(len(mod.limit_branches)):
                lim_sf += 1
                vold.append(xat(len(seen_nodes) + cur_sf + q_sf + lim_sf - 1))
            cur_sf += sum(_shorts_connected(mod, shorts_on, idx))   # potential contributions with a branch current (vasim.jl:2365, 3253-3280)
            mf = float(np.asarray(resolve(d.params["m"], params)).flat[0])
            vals = va.host_eval.evaluate(mod, V, par, 27.0 + 273.15, mf, 1e-12, vold=vold, given=set(d.model))
            flags, pos = [], 0
            for b, (pn, nn) in enumerate(mod.branches):
                if not mod.reactive[b] or not active[b]:
                    flags.append(False)
                    continue
                Vb = (V[pn] if pn >= 0 else 0.0) - (V[nn] if nn >= 0 else 0.0)
                Q = mf * vals[b][1]
                if pos >= len(Qs):
                    is_vdep.append(False); Qs.append(Q); Vs.append(Vb)
                else:
                    if abs(Vb) > 1e-6 and abs(Vs[pos]) > 1e-6:
                        Cc, Cs = Q / Vb, Qs[pos] / Vs[pos]
                        diff, mx = abs(Cc - Cs), max(abs(Cc), abs(Cs))
                        if diff > 1e-15 and (mx < 1e-30 or diff / mx > 1e-6):
                            is_vdep[pos] = True
                    Qs[pos], Vs[pos] = Q, Vb
                flags.append(is_vdep[pos])
                q_sf += int(is_vdep[pos])
                pos += 1
            last[d.name] = tuple(flags)
        n_q_prev = q_sf - n_m1q
    return last


def discover(circuit: Circuit, params: Dict[str, float]) -> Structure:
    """Structure of ``circuit`` for parameter set ``params`` (first sweep instance; every
    instance of a batch must share it)."""
    A = _Alloc()
    wave_data: List[float] = []
    per_type: Dict[str, list] = {}
    order: List[str] = []
    breakpoints = []
    vdep = detect_mos1_vdep(circuit, params)
    va_vdep = detect_va_vdep(circuit, params)
    pending_bsrc = []   # (ipar list, tokens): programs are encoded once every node has its index
    opinfo = []
    recs = []   # (stream, seq-order implicit, type, dev_in_block, local_slot, row_typed, col_typed)
    for di, dev in enumerate(circuit.devices):
        ty = dev.type
        if ty not in per_type:
            per_type[ty] = []
            order.append(ty)
        nodes = [A.node(nm) for nm in dev.nodes]
        ipar = []
        prog = PROGRAMS.get(ty)
        if ty in ("L", "V", "BV"):
            nodes.append(A.current("I_" + dev.name))
        elif ty == "E":
            nodes.append(A.current("I_" + dev.name))
        elif ty == "H":
            nodes.append(A.current("I_" + dev.name + "_in"))
            nodes.append(A.current("I_" + dev.name + "_out"))
        elif ty == "F":
            nodes.append(A.current("I_" + dev.name + "_in"))
        if ty in ("V", "I"):
            ipar = list(_wave_ipar(dev.wave, wave_data))
            bp = wave_breakpoints(dev.wave)
            if bp is not None:
                breakpoints.append(bp)
        if ty in ("BV", "BI"):
            ipar = [0, 0]
            tokens = bsource.compile_expr(dev.params["expr"])
            for tok in tokens:           # get_voltage(name) touches its nodes in evaluation order
                if tok[0] == "v":
                    A.node(tok[1]); A.node(tok[2])
            pending_bsrc.append((ipar, tokens))
        if ty == "D":
            lim = bool(dev.params.get("limit", True))
            ipar = [1 if lim else 0]
            if lim:
                Is = float(resolve(dev.params["Is"], params))
                nVt = float(resolve(dev.params["n"], params)) * float(resolve(dev.params["Vt"], params))
                vcrit = nVt * np.log(nVt / (np.sqrt(2.0) * Is))
                nodes.append(A.limit(dev.name + "_vdlim", vcrit))
            else:
                nodes.append(GND)
                prog = prog[3:]
        if ty == "MOS1":
            given = {k: resolve(v, params) for k, v in dev.model.items()}
            sc_d, sc_s = m1.short_circuits(given)
            d_int = nodes[0] if sc_d else A.node("%s_sp_mos1_d_int" % dev.name)
            s_int = nodes[2] if sc_s else A.node("%s_sp_mos1_s_int" % dev.name)
            nodes += [d_int, s_int]
            names = ("g_s_int", "d_int_s_int", "b_s_int", "b_d_int")
            nodes += [A.limit("%s_sp_mos1_lim_%s" % (dev.name, nm), 0.0) for nm in names]
            qn = ("g", "b", "d_int", "s_int")
            qs = []
            # charges are allocated while the branches are stamped, in branch order
            for r in range(4):
                qs.append(A.charge("%s_sp_mos1_Q_%s_0" % (dev.name, qn[r])) if vdep[r] else GND)
            nodes += qs
            ipar = [sum((1 << r) for r in range(4) if vdep[r])]
            prog = mos1_program(vdep)
        if ty.startswith("VA:"):
            mid, mod = va.get(ty[3:])
            if len(dev.nodes) != len(mod.ports):
                raise ValueError("%s: %d nets for the %d ports of %s" % (dev.name, len(dev.nodes), len(mod.ports), mod.name))
            # internal nodes, then one charge unknown per voltage-dependent reactive branch, allocated in branch order as
            # the branches are stamped (vasim.jl:3533-3564, 3433-3472)
            _, alias, shorts_on, active = va_instance(mod, dev, params)
            for k in range(len(mod.ports), mod.n_nodes):     # a collapsed internal node is its neighbour's unknown (vasim.jl:3533-3564)
                if k in alias and alias[k] < 0:
                    nodes.append(GND)                         # V(a) <+ 0: the internal node is ground
                else:
                    nodes.append(nodes[alias[k]] if k in alias and alias[k] < k else None if k in alias else A.node("%s_%s_%s" % (dev.name, mod.name, mod.nodes[k])))
            for k in range(len(mod.ports), mod.n_nodes):     # ... also when the neighbour is a later internal node
                if nodes[k] is None:
                    nodes[k] = nodes[alias[k]]
            vd = va_vdep[dev.name]
            for b, (p, n) in enumerate(mod.branches):
                nm = "%s_%s_Q_%s_%s" % (dev.name, mod.name, mod.nodes[p] if p >= 0 else "0", mod.nodes[n] if n >= 0 else "0")
                nodes.append(A.charge(nm) if (mod.reactive[b] and vd[b]) else GND)
            for (p, n) in mod.limit_branches:      # one limit unknown per $limit probe branch, init 0 (vasim.jl:3110-3138)
                nodes.append(A.limit("%s_%s_lim_%s_%s" % (dev.name, mod.name, mod.nodes[p] if p >= 0 else "0", mod.nodes[n] if n >= 0 else "0"), 0.0))
            shorts_on = _shorts_connected(mod, shorts_on, nodes)
            for j, si in enumerate(mod.vshorts):   # the branch current of an executed potential contribution (vasim.jl:2365: I_V_<p>_<n>; 3262, 3274)
                nodes.append(A.current(mod.short_current_name(j, dev.name)) if shorts_on[j] else GND)
            ipar = [mid, sum((1 << b) for b in range(len(mod.branches)) if mod.reactive[b] and vd[b]), sum((1 << j) for j in range(len(mod.vshorts)) if shorts_on[j])]
            prog = mod.program(vd, active, shorts_on)
        d_in_block = len(per_type[ty])
        per_type[ty].append((di, nodes, ipar))
        # operating-point channel (context.jl:1200-1342): the device's terminals and, per terminal, the local KCL rows whose
        # contributions flow into it (an internal node collapsed onto a terminal feeds that terminal)
        if ty == "MOS1":
            terms, groups = ("d", "g", "s", "b"), [[0] + ([4] if sc_d else []), [1], [2] + ([5] if sc_s else []), [3]]
        elif ty.startswith("VA:"):
            terms = tuple(p.lower() for p in mod.ports)
            groups = [[t] + [k for k in alias if alias[k] == t] for t in range(len(mod.ports))]
        elif ty == "D":
            terms, groups = ("a", "c"), [[0], [1]]
        elif ty == "SMOS":
            terms, groups = ("d", "g", "s"), [[0], [1], [2]]
        else:
            terms, groups = ("p", "n"), [[0], [1]]
        opinfo.append({"name": dev.name, "type": ty, "dev": d_in_block, "nodes": list(nodes), "prog": list(prog), "terminals": terms, "groups": groups})
        for (stream, k, rl, cl) in prog:
            row = nodes[rl]
            col = nodes[cl] if cl is not None else None
            if row == GND or (cl is not None and col == GND):
                continue      # ground skipped before the position counter (value_only.jl:395-397)
            recs.append((stream, ty, d_in_block, k, row, col))
    n_nodes, n_cur, n_q, n_l = len(A.node_names), len(A.current_names), len(A.charge_names), len(A.limit_names)
    n = n_nodes + n_cur + n_q + n_l
    for ipar, tokens in pending_bsrc:
        def node_index(nm):
            if nm in ("0", "gnd", "gnd!"):
                return -1
            return A.node_idx[nm]      # KeyError: the expression names a node no device connects to
        prog = bsource.encode(tokens, node_index)
        ipar[0], ipar[1] = len(wave_data), len(prog)
        wave_data.extend(prog)
    # blocks
    blocks: List[Block] = []
    gb = cb = bb = 0
    block_of = {}
    for ty in order:
        items = per_type[ty]
        cnt = len(items)
        nl, ng, nc, nb, npar, nip = shape_of(ty)
        nodes = np.full((nl, cnt), -1, dtype=np.int32)
        ipar = np.zeros((max(nip, 1), cnt), dtype=np.int32)
        for j, (di, nd, ip) in enumerate(items):
            for k, t in enumerate(nd):
                nodes[k, j] = A.resolve(t)
            for k, v in enumerate(ip):
                ipar[k, j] = v
        blk = Block(ty, cnt, [it[0] for it in items], nodes, ipar, gb, cb, bb, ng, nc, nb, npar)
        block_of[ty] = blk
        blocks.append(blk)
        gb += ng * cnt
        cb += nc * cnt
        bb += nb * cnt
    ns_g, ns_c, ns_b = gb, cb, bb
    # COO streams -> pattern + gather lists
    coo = {"G": [], "C": [], "b": []}
    for (stream, ty, dj, k, row, col) in recs:
        blk = block_of[ty]
        base = {"G": blk.g_base, "C": blk.c_base, "b": blk.b_base}[stream]
        slot = base + k * blk.count + dj
        coo[stream].append((slot, A.resolve(row), A.resolve(col) if col is not None else -1))
    ent = sorted(set((r, c) for (_, r, c) in coo["G"]) | set((r, c) for (_, r, c) in coo["C"]))
    nnz = len(ent)
    rows = np.array([e[0] for e in ent], dtype=np.int64)
    cols = np.array([e[1] for e in ent], dtype=np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    colidx = cols.astype(np.int32)
    pos = {e: k for k, e in enumerate(ent)}
    # reference CSC order: sorted by (col, row)
    csc_order = np.lexsort((rows, cols))
    to_ref = np.empty(nnz, dtype=np.int32)
    to_ref[csc_order] = np.arange(nnz, dtype=np.int32)
    ref_colptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(ref_colptr, cols + 1, 1)
    ref_colptr = np.cumsum(ref_colptr).astype(np.int32)
    ref_rowval = rows[csc_order].astype(np.int32)

    def gather(stream, nout, key):
        lists = [[] for _ in range(nout)]
        for (slot, r, c) in coo[stream]:     # already in COO (stamp) order
            lists[key(r, c)].append(slot)
        ptr = np.zeros(nout + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([len(l) for l in lists])
        flat = np.array([s for l in lists for s in l], dtype=np.int32)
        return ptr, flat

    g_ptr, g_slots = gather("G", nnz, lambda r, c: pos[(r, c)])
    c_ptr, c_slots = gather("C", nnz, lambda r, c: pos[(r, c)])
    b_ptr, b_slots = gather("b", n, lambda r, c: r)
    diag = np.array([pos.get((i, i), -1) for i in range(n_nodes)], dtype=np.int32)
    return Structure(
        n=n, n_nodes=n_nodes, n_currents=n_cur, n_charges=n_q, n_limits=n_l,
        node_names=A.node_names, current_names=A.current_names, charge_names=A.charge_names,
        limit_names=A.limit_names, rowptr=rowptr, colidx=colidx, to_ref_nz=to_ref,
        ref_colptr=ref_colptr, ref_rowval=ref_rowval, blocks=blocks,
        wave_data=np.array(wave_data, dtype=np.float64), ns_g=ns_g, ns_c=ns_c, ns_b=ns_b,
        g_ptr=g_ptr, g_slots=g_slots, c_ptr=c_ptr, c_slots=c_slots, b_ptr=b_ptr, b_slots=b_slots,
        diag_nz=diag, limit_init=np.array(A.limit_init, dtype=np.float64), breakpoints=breakpoints,
        n_coo_g=len(coo["G"]), n_coo_c=len(coo["C"]), n_coo_b=len(coo["b"]), mos1_vdep=vdep, opinfo=opinfo)


def pack_params(st: Structure, circuit: Circuit, params: Dict[str, np.ndarray], temp_c, B: int,
                gmin: float = 1e-12, tnom_c: float = 27.0) -> List[np.ndarray]:
    """Per-instance parameter blocks, one array [B, n_par, count] per device block.
    ``params``: sweepable circuit parameters, each a scalar or a [B] array; ``temp_c``: [B] or scalar."""
    out = []
    for blk in st.blocks:
        arr = np.zeros((B, blk.n_par, blk.count))
        for j, di in enumerate(blk.dev_index):
            dev = circuit.devices[di]
            g = lambda k: np.asarray(resolve(dev.params[k], params), dtype=float)
            ty = blk.type
            if ty == "R":
                arr[:, 0, j] = 1.0 / g("r")
            elif ty == "C":
                arr[:, 0, j] = g("c")
            elif ty == "L":
                arr[:, 0, j] = g("l")
            elif ty in ("V", "I"):
                arr[:, 0, j] = g("dc")
                arr[:, 1, j] = g("scale")
            elif ty in ("BV", "BI"):
                arr[:, 0, j] = g("scale")
            elif ty in ("E", "F"):
                arr[:, 0, j] = g("gain")
            elif ty == "G":
                arr[:, 0, j] = g("gm")
            elif ty == "H":
                arr[:, 0, j] = g("rm")
            elif ty == "D":
                Is, nVt = g("Is"), g("n") * g("Vt")
                arr[:, 0, j] = Is
                arr[:, 1, j] = nVt
                arr[:, 2, j] = nVt * np.log(nVt / (np.sqrt(2.0) * Is))   # vcrit devices.jl:1319-1320
            elif ty == "DCAP":
                arr[:, 0, j] = g("Is")
                arr[:, 1, j] = g("n") * g("Vt")
                arr[:, 2, j] = g("Cj0")
                arr[:, 3, j] = g("Vj")
                arr[:, 4, j] = g("m")
            elif ty == "SMOS":
                for k, nm in enumerate(("Vth", "K", "lambda", "Cgd", "Cgs")):
                    arr[:, k, j] = g(nm)
            elif ty == "MOS1":
                given = {k: resolve(v, params) for k, v in dev.model.items()}
                der, _ = m1.derive(given, temp_c, tnom_c, gmin, mfactor=g("m"))
                arr[:, :, j] = der.T if der.shape[1] == B else np.repeat(der.T, B, axis=0)
            elif ty.startswith("VA:"):
                mod = va.get(ty[3:])[1]
                par = va.host_eval.defaults(mod, {k: resolve(v, params) for k, v in dev.model.items()})
                for k, nm in enumerate(mod.params):
                    if mod.param_kind.get(nm) != "string":
                        arr[:, k, j] = par[nm]
                np_ = len(mod.params)
                for k, (sp, lit) in enumerate(mod.string_tests):               # string parameter == literal: decided here
                    arr[:, np_ + 3 + (np_ if mod.uses_given else 0) + k, j] = 1.0 if str(par[sp]) == lit else 0.0
                for k, call in enumerate(mod.table_calls):                     # $table_model of parameters: evaluated here, per instance
                    base = np_ + 3 + (np_ if mod.uses_given else 0) + len(mod.string_tests) + k
                    for i in range(B):
                        pi = {nm: (float(np.asarray(v).flat[i if np.size(v) > 1 else 0]) if not isinstance(v, str) else v) for nm, v in par.items()}
                        arr[i, base, j] = va.host_eval.table_value(mod, call, [va.host_eval.static_eval(a, pi) for a in call[2][:-2]])
                arr[:, np_, j] = np.asarray(temp_c, dtype=float) + 273.15     # $temperature
                arr[:, np_ + 1, j] = g("m")                                     # $mfactor
                arr[:, np_ + 2, j] = gmin                                       # $simparam("gmin")
                if mod.uses_given:                                              # $param_given: one flag per parameter
                    given = {k.lower() for k in dev.model} | {mod.aliasparams[k].lower() for k in dev.model if k in mod.aliasparams}
                    for k, nm in enumerate(mod.params):
                        arr[:, np_ + 3 + k, j] = 1.0 if nm.lower() in given else 0.0
        out.append(np.ascontiguousarray(arr))
    return out
