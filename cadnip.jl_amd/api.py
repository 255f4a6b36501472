"""Reference-shaped user API on top of the GPU hot path.

Mirrors the surface a Cadnip.jl user touches on this path -- same names, argument meaning and
result layout -- so that tests read like the reference's own:

    MNASpec, MNACircuit, alter                       /root/reference/src/mna/solve.jl:57-70, 1585-1597, 1719-1732
    dc(circuit) / tran(circuit, tspan)               /root/reference/src/sweeps.jl:450-455, 588-665   (dc! / tran!)
    Sweep, ProductSweep, TandemSweep, SerialSweep    /root/reference/src/sweeps.jl:150-330
    CircuitSweep, SweepResult, dc(cs), tran(cs)      /root/reference/src/sweeps.jl:387-424, 477-532, 692-707
    solution layout [V | I | q*1e12 | v_lim], sol[name]   /root/reference/src/mna/build.jl:39-51, 421-457

Differences forced by the target: a ``MNACircuit`` wraps a flattened device table instead of a
generated Julia builder; ``tran``/``dc`` over a ``CircuitSweep`` integrate all sweep points as
one resident batch on the GPU instead of the reference's serial loop (sweeps.jl:696-703);
temperature is a per-point axis (``temp`` key), which the reference can only reach through an
outer loop over ``MNASpec``.
"""
import itertools
from dataclasses import dataclass, field, replace
from typing import Any, Dict, List, Optional

import numpy as np

from .circuit import Circuit
from .structure import Structure, discover, expand_breakpoints, pack_params


@dataclass(frozen=True)
class MNASpec:
    temp: float = 27.0
    mode: str = "tran"
    time: float = 0.0
    gmin: float = 1e-12
    gshunt: float = 0.0
    srcFact: float = 1.0
    tnom: float = 27.0
    abstol: float = 1e-12
    reltol: float = 1e-3
    vntol: float = 1e-6
    iabstol: float = 1e-12


def with_mode(spec, mode):
    return replace(spec, mode=mode)


def with_temp(spec, temp):
    return replace(spec, temp=float(temp))


@dataclass(frozen=True)
class MNACircuit:
    circuit: Circuit
    params: Dict[str, float] = field(default_factory=dict)
    spec: MNASpec = MNASpec()


def alter(c: MNACircuit, spec: Optional[MNASpec] = None, **params) -> MNACircuit:
    """solve.jl:1719-1732: new circuit with some parameters (or the spec) replaced."""
    p = dict(c.params)
    for k, v in params.items():
        if k not in p:
            raise KeyError("unknown circuit parameter %r" % k)
        p[k] = v
    return MNACircuit(c.circuit, p, spec if spec is not None else c.spec)


# ------------------------------------------------------------------------------------------------
# sweeps (sweeps.jl:150-330)
# ------------------------------------------------------------------------------------------------
class Sweep:
    def __init__(self, **axes):
        if len(axes) != 1:
            raise ValueError("Sweep takes exactly one name=values axis; combine with ProductSweep/TandemSweep")
        (self.name, vals), = axes.items()
        self.values = list(vals)

    def __iter__(self):
        return ({self.name: v} for v in self.values)

    def __len__(self):
        return len(self.values)


class ProductSweep:
    """Cartesian product, first axis fastest (Base.Iterators.product, sweeps.jl:272)."""

    def __init__(self, *sweeps, **axes):
        self.sweeps = list(sweeps) + [Sweep(**{k: v}) for k, v in axes.items()]

    def __iter__(self):
        lists = [list(s) for s in self.sweeps]
        for combo in itertools.product(*reversed(lists)):
            d = {}
            for part in reversed(combo):
                d.update(part)
            yield d

    def __len__(self):
        return int(np.prod([len(s) for s in self.sweeps]))


class TandemSweep:
    def __init__(self, *sweeps, **axes):
        self.sweeps = list(sweeps) + [Sweep(**{k: v}) for k, v in axes.items()]
        if len({len(s) for s in self.sweeps}) > 1:
            raise ValueError("TandemSweep axes must have equal lengths")

    def __iter__(self):
        for parts in zip(*self.sweeps):
            d = {}
            for p in parts:
                d.update(p)
            yield d

    def __len__(self):
        return len(self.sweeps[0]) if self.sweeps else 0


class SerialSweep:
    def __init__(self, *sweeps):
        self.sweeps = list(sweeps)

    def __iter__(self):
        return itertools.chain(*self.sweeps)

    def __len__(self):
        return sum(len(s) for s in self.sweeps)


class CircuitSweep:
    """sweeps.jl:387-424.  Sweep keys name circuit parameters; the extra key ``temp`` sweeps
    MNASpec.temp."""

    def __init__(self, circuit: MNACircuit, iterator):
        self.circuit = circuit
        self.iterator = iterator
        for pt in iterator:
            for k in pt:
                if k != "temp" and k not in circuit.params:
                    raise KeyError("sweep variable %r is not a circuit parameter" % k)

    def points(self):
        return list(self.iterator)

    def __len__(self):
        return len(self.iterator)


class SweepResult:
    """sweeps.jl:477-487: iterates (params, solution) pairs."""

    def __init__(self, points, solutions):
        self.points = points
        self.solutions = solutions

    def __iter__(self):
        return iter(zip(self.points, self.solutions))

    def __len__(self):
        return len(self.points)

    def __getitem__(self, i):
        return self.solutions[i]


# ------------------------------------------------------------------------------------------------
# solutions
# ------------------------------------------------------------------------------------------------
class DCSolution:
    """solve.jl:156-166; name lookup in the reference's order -- nodes, currents, charges, limits, then device terminal
    currents ``i_<device>_<terminal>`` and operating-point variables ``<device>_<var>`` (solve.jl:234-250, test/opinfo.jl)."""

    def __init__(self, st: Structure, x, converged, op=None):
        self.st = st
        self.x = np.asarray(x)
        self.converged = bool(converged)
        self.op = dict(op or {})
        self.node_names = st.node_names
        self.current_names = st.current_names
        self.charge_names = st.charge_names
        self.limit_names = st.limit_names
        self.n_nodes = st.n_nodes

    def __getitem__(self, name):
        try:
            return float(self.x[self.st.index_of(name)])
        except KeyError:
            if name in self.op:
                return self.op[name]
            raise

    def terminal_currents(self):
        return {k: v for k, v in self.op.items() if k.startswith("i_")}

    def op_vars(self):
        return {k: v for k, v in self.op.items() if not k.startswith("i_")}

    def keys(self):
        return list(self.st.node_names) + list(self.st.current_names) + list(self.op)

    def __contains__(self, name):
        return name in self.op or name in self.st.node_names or name in self.st.current_names


class TranSolution:
    """What the path hands back: states at the requested ``saveat`` times plus solver statistics
    (sol.t, sol[name], sol(t), sol.stats.nnonliniter as in benchmarks/vacask/ring/cedarsim/runme.jl:74-76)."""

    def __init__(self, st, t, u, stats, retcode):
        self.st = st
        self.t = np.asarray(t)
        self.u = np.asarray(u)          # [n_save, n]
        self.stats = stats
        self.retcode = retcode

    def __getitem__(self, name):
        return self.u[:, self.st.index_of(name)]

    def __call__(self, t, name=None):
        cols = self.u if name is None else self.u[:, self.st.index_of(name)]
        if cols.ndim == 1:
            return float(np.interp(t, self.t, cols))
        return np.array([np.interp(t, self.t, cols[:, j]) for j in range(cols.shape[1])])


def nameat(sol, name, t):
    """solve.jl:347-353"""
    return sol(t, name)


# ------------------------------------------------------------------------------------------------
# batched simulator
# ------------------------------------------------------------------------------------------------
def clip_sample(J, headroom=1e6):
    """|J| of one or more sample Jacobians ([.., nnz]) for the symbolic phase, each sample clipped to ``headroom`` times
    the median of its non-zero magnitudes (non-finite entries count as the cap).  A random probe point can forward-bias
    an exponential junction by volts: its conductance (1e30 and more) would swamp every other entry of the sample, and
    the pivot search -- which eliminates numerically -- would see a singular matrix where only the probe was absurd."""
    a = np.abs(np.atleast_2d(np.asarray(J, dtype=float)))
    out = np.empty_like(a)
    for i, row in enumerate(a):
        fin = np.isfinite(row)
        nz = row[fin & (row > 0)]
        cap = headroom * float(np.median(nz)) if nz.size else 0.0
        r = np.where(fin, row, cap)
        out[i] = np.minimum(r, cap) if nz.size else 0.0
    return out


class BatchSimulator:
    """One structure, B resident sweep instances on one GPU."""

    @classmethod
    def from_packed(cls, st: Structure, packed, spec: Optional[MNASpec] = None, device: int = 0, vscale: float = 1.0):
        """A simulator from a ready structure and packed per-instance parameter blocks (``pack_params`` output, one array
        [B, n_par, count] per block) -- e.g. loaded with ``structure.load_structure`` on a machine that holds the library's generated
        model code but not the model's Verilog-A source."""
        from . import hip
        self = cls.__new__(cls)
        spec = spec or MNASpec()
        self.mc, self.points, self.st = None, None, st
        self.B = int(np.asarray(packed[0]).shape[0])
        self.params, self.temps = {}, np.full(self.B, spec.temp)
        self._vscale = float(vscale)
        self.h = hip.Handle(st, self.B, device)
        self.h.set_params([np.ascontiguousarray(p, dtype=np.float64) for p in packed])
        self.h.set_spec(mode=spec.mode if spec.mode in ("dcop", "tran", "tranop") else "tran", gmin=spec.gmin, gshunt=spec.gshunt, srcFact=spec.srcFact)
        self._analyzed = False
        return self

    def __init__(self, mc: MNACircuit, points: Optional[List[Dict[str, Any]]] = None, device: int = 0, st: Optional[Structure] = None):
        from . import hip
        self.mc = mc
        points = points if points else [{}]
        self.points = points
        B = len(points)
        self.B = B
        params = {k: np.full(B, float(v)) for k, v in mc.params.items()}
        temps = np.full(B, mc.spec.temp)
        for i, pt in enumerate(points):
            for k, v in pt.items():
                if k == "temp":
                    temps[i] = float(v)
                else:
                    params[k][i] = float(v)
        self.params, self.temps = params, temps
        p0 = {k: float(v[0]) for k, v in params.items()}
        self.st = st if st is not None else discover(mc.circuit, p0)
        self.h = hip.Handle(self.st, B, device)
        self.h.set_params(pack_params(self.st, mc.circuit, params, temps, B, gmin=mc.spec.gmin, tnom_c=mc.spec.tnom))
        self.h.set_spec(mode=mc.spec.mode if mc.spec.mode in ("dcop", "tran", "tranop") else "tran",
                        gmin=mc.spec.gmin, gshunt=mc.spec.gshunt, srcFact=mc.spec.srcFact)
        self._analyzed = False

    def close(self):
        self.h.close()

    def vscale(self):
        if self.mc is None:
            return self._vscale
        v = [abs(float(np.max(np.abs(self.params[k])))) for k in self.params] + [1.0]
        for d in self.mc.circuit.devices:
            if d.type == "V":
                dc = d.params.get("dc", 0.0)
                if not hasattr(dc, "name"):
                    v.append(abs(float(dc)))
                if d.wave is not None and d.wave[0] == "pwl":
                    v.append(max(abs(float(y)) for y in d.wave[2]))
        return max(v)

    def analyze(self, gamma=1e9, n_samples=6, seed=1234, sample=None):
        """Symbolic LU phase on the element-wise max |G + gamma*C| over several operating points
        (cold start with initjct, zero, and random points) and over ALL instances of the handle, so the static pivot order suits all.
        Each sample is clipped (``clip_sample``) before it enters the max.  The order therefore belongs to the batch: a subset of
        its points analysed on its own may get another one and then agrees with the batch to rounding, not bit for bit;
        ``sample`` (the ``pivot_sample`` of another simulator of the same structure) reuses that simulator's order."""
        if sample is not None:
            self.pivot_sample = np.array(sample, dtype=float)
            self.h.analyze_values(self.pivot_sample)
            self._analyzed = True
            return
        rng = np.random.default_rng(seed)
        st, h = self.st, self.h
        vs = self.vscale()
        acc = np.zeros(st.nnz)
        for k in range(n_samples):
            if k == 0:
                u = np.zeros(st.n)
                u[st.n - st.n_limits:] = st.limit_init
                h.set_initjct(True)
            elif k == 1:
                u = np.zeros(st.n)
            else:
                u = (rng.random(st.n) * 1.2 - 0.1) * vs
                u[st.n_nodes:st.n_nodes + st.n_currents] = 0.0
            h.rebuild(u, 0.0)
            h.set_initjct(False)
            J = h.jacobian(gamma)
            acc = np.maximum(acc, np.max(clip_sample(J), axis=0))
        h.analyze_values(acc)
        self.pivot_sample = acc
        self._analyzed = True

    def analyze_at(self, u, t=0.0, gamma=0.0, instance=0):
        """Symbolic LU phase on the Jacobian ``G + gamma*C`` of one state -- KLU's first ``klu_factor``: the pivot order of
        the matrix the solver is about to meet (e.g. the first transient steps from a given start state, gamma = 1 / h).
        For circuits whose composite sample (``analyze``) does not yield a usable order."""
        uu = np.broadcast_to(np.asarray(u, dtype=float), (self.B, self.st.n)) if np.ndim(u) == 1 else np.asarray(u, dtype=float)
        self.h.rebuild(uu, t)
        self.h.analyze_values(self.h.jacobian(gamma)[instance])
        self._analyzed = True

    def dc(self, u0=None, abstol=1e-10, maxiters=100, mode="dcop", fused=False, participate=None, cold_start=None):
        self.h.set_spec(mode=mode)
        if not self._analyzed:
            self.analyze()
        return self.h.dc_run(u0, abstol=abstol, maxiters=maxiters, use_pcnr=True, cold_start=(u0 is None) if cold_start is None else cold_start,
                             fused=fused, participate=participate)

    def operating_points(self, u, mode="dcop"):
        """Terminal currents and op variables of every instance at the states ``u`` (opinfo.py): one restamp with the
        per-device contributions read back."""
        from . import opinfo
        self.h.set_spec(mode=mode)
        self.h.set_u(u)
        Sg, Sc, Sb = self.h.get_contributions()
        packed = pack_params(self.st, self.mc.circuit, self.params, self.temps, self.B, gmin=self.mc.spec.gmin, tnom_c=self.mc.spec.tnom)
        return [opinfo.operating_point(self.st, np.asarray(u)[i], Sg[i], Sb[i], packed, i) for i in range(self.B)]

    def dc_continuation(self, abstol=1e-10, maxiters=100, mode="dcop", fused=False, serial=False):
        """dc!(cs; continuation=true) (sweeps.jl:489-532) for a resident batch.  The reference walks the sweep serially and
        starts every point from the last CONVERGED solution.  Here the sweep is solved in 1 + ceil(log2 B) batch stages
        (``continuation_stages``): point 0 cold, then the midpoints, quarter points, ... each started from the nearest
        converged point of the earlier stages at a lower index (the reference's direction), else the nearest converged
        one at all, else cold; a point that failed is never a starting guess (sweeps.jl:522-524).  For a circuit with ONE
        operating point continuation changes the path Newton takes, not where it lands (test/sweep.jl:332-345).  A circuit with
        several (a latch, a Schmitt trigger, the flip-flop itself) is different: the reference's strictly serial chain i - 1 -> i
        follows one branch of the hysteresis, while a seed from B / 2 points away can land on another.  ``serial=True`` is that
        chain: B stages of one point each, every point started from its converged predecessor (B launches instead of log2 B --
        the price of following a branch).  Returns (u, converged, stats)."""
        st = self.st
        u = np.zeros((self.B, st.n))
        conv = np.zeros(self.B, dtype=bool)
        solved = np.zeros(self.B, dtype=bool)
        total = {"newton_iters": 0, "stages": 0, "cold_points": 0}
        for stage in ([[i] for i in range(self.B)] if serial else continuation_stages(self.B)):
            start = np.zeros((self.B, st.n))
            mask = np.zeros(self.B, dtype=bool)
            mask[stage] = True
            good = np.flatnonzero(solved & conv)
            for i in stage:
                j = seed_for(i, good)
                if j is None:
                    total["cold_points"] += 1                    # zeros: dc_run seeds the limit variables and arms initjct for it
                else:
                    start[i] = u[j]
            start[~mask] = u[~mask]
            ui, ci, stats = self.dc(start, abstol=abstol, maxiters=maxiters, mode=mode, fused=fused, participate=mask, cold_start=True)
            u[mask], conv[mask] = ui[mask], ci[mask]
            solved |= mask
            total["newton_iters"] += stats["newton_iters"]
            total["stages"] += 1
        return u, conv, total

    def tran(self, tspan, abstol, reltol, saveat, initializealg="tranop", u0=None, warmup_dt=1e-12, **kw):
        """``initializealg``: "tranop" = CedarTranOp, a DC solve in :tranop mode at t0 (dcop.jl:160-212); "uic" = CedarUICOp
        (dcop.jl:109-151, 304-411): no DC solve -- the run starts from ``u0`` (zeros by default) and the integrator's first
        steps, backward Euler from ``warmup_dt``, relax the algebraic constraints (for oscillators and for circuits whose
        static operating point Newton does not find)."""
        st = self.st
        if initializealg not in ("tranop", "uic"):
            raise ValueError("initializealg must be 'tranop' or 'uic'")
        breaks = expand_breakpoints(st.breakpoints, tspan)
        if initializealg == "uic":
            if isinstance(u0, dict):          # {unknown name: value}: the usual .IC form
                start = np.zeros((self.B, st.n))
                for name, value in u0.items():
                    start[:, st.index_of(name)] = value
            else:
                start = np.zeros((self.B, st.n)) if u0 is None else np.broadcast_to(np.asarray(u0, dtype=float), (self.B, st.n)).copy()
            if not self._analyzed:     # pivot order from the Jacobian the first steps will meet (h = warmup_dt)
                self.analyze_at(start, t=tspan[0], gamma=1.0 / (10.0 * warmup_dt))
            self.h.set_u(start)
            self.h.set_spec(mode="tran")
            kw.setdefault("h0", warmup_dt)
            out, per, stats = self.h.tran_run(tspan[0], tspan[1], abstol, reltol, breaks=breaks, save_t=saveat, **kw)
            stats["dc_newton_iters"] = 0
            return out, per, stats
        if not self._analyzed:
            self.analyze()
        u0_, conv, dcs = self.dc(abstol=1e-9, mode="tranop", fused=bool(kw.get("fused", False)))
        if not np.all(conv):
            raise RuntimeError("transient initialisation (CedarTranOp) failed for %d instance(s)" % int((~conv).sum()))
        self.h.set_spec(mode="tran")
        out, per, stats = self.h.tran_run(tspan[0], tspan[1], abstol, reltol, breaks=breaks, save_t=saveat, **kw)
        stats["dc_newton_iters"] = dcs["newton_iters"]
        return out, per, stats


def continuation_stages(n):
    """Index sets of the staged continuation: [0], then for stride s = 2^k (largest first) the indices i = s (mod 2s): each
    lies midway between two indices of the earlier stages, and i - s is always among them."""
    if n <= 0:
        return []
    stages = [[0]]
    s = 1
    while s < n:
        s *= 2
    s //= 2
    while s >= 1:
        idx = list(range(s, n, 2 * s))
        if idx:
            stages.append(idx)
        s //= 2
    return stages


def seed_for(i, good):
    """the converged point a staged-continuation point starts from: the nearest one below it, else the nearest at all"""
    good = np.asarray(good)
    if good.size == 0:
        return None
    below = good[good < i]
    if below.size:
        return int(below[-1])
    return int(good[np.argmin(np.abs(good - i))])


def structure_classes(mc: MNACircuit, points):
    """Partition sweep points by circuit STRUCTURE (unknowns, pattern, slot maps): a sweep may move a parameter across a
    value that changes it -- a series resistance reaching zero collapses an internal node (mos1.va:716-721,
    vasim.jl:3537-3553), a junction capacitance reaching zero removes a charge unknown (contrib.jl:214-257).  The reference
    rebuilds every point from scratch; here every class gets its own resident batch.  Discovery runs once per distinct
    value tuple of the parameters that device MODEL cards refer to (sweeping a source value or the temperature never
    changes the structure).  Returns [(point indices, Structure)]."""
    from .circuit import Param
    model_pars = sorted({v.name for d in mc.circuit.devices if d.model for v in d.model.values() if isinstance(v, Param)})
    classes, by_key, sig_class = [], {}, {}
    for i, pt in enumerate(points):
        p = dict(mc.params)
        p.update({k: v for k, v in pt.items() if k != "temp"})
        key = tuple(float(p[k]) for k in model_pars)
        if key not in by_key:
            st = discover(mc.circuit, {k: float(v) for k, v in p.items()})
            sig = st.signature()
            if sig not in sig_class:
                sig_class[sig] = len(classes)
                classes.append(([], st))
            by_key[key] = sig_class[sig]
        classes[by_key[key]][0].append(i)
    return classes


def _resolve_abstol(abstol, st):
    """_resolve_abstol (sweeps.jl:626): per-class NamedTuple -> vector via state_abstol."""
    if isinstance(abstol, dict):
        return st.state_abstol(**abstol)
    return np.broadcast_to(np.asarray(abstol, dtype=float), (st.n,)).copy()


def dc(target, u0=None, device=0, continuation=True):
    """dc!(circuit) / dc!(cs::CircuitSweep; continuation=true) -- sweeps.jl:450-455, 489-532.  Goes through
    with_mode(:dcop), which keeps only temp+mode of the spec (solve.jl:1976-1989).  A sweep is partitioned by structure
    (``structure_classes``); each class is one resident batch, solved with the staged continuation (``continuation=True``), with the
    reference's strictly serial chain point i - 1 -> point i (``continuation="serial"``: the choice for circuits with several DC
    solutions, whose branch a sweep is meant to follow -- see ``BatchSimulator.dc_continuation``), or as independent cold solves
    (``continuation=False``)."""
    if isinstance(target, CircuitSweep):
        mc = target.circuit
        mc = MNACircuit(mc.circuit, mc.params, MNASpec(temp=mc.spec.temp, mode="dcop"))
        pts = target.points()
        sols = [None] * len(pts)
        for idx, st in structure_classes(mc, pts):
            sim = BatchSimulator(mc, [pts[i] for i in idx], device, st=st)
            try:
                u, conv, _ = sim.dc_continuation(serial=continuation == "serial") if continuation else sim.dc()
                ops = sim.operating_points(u)
                for k, i in enumerate(idx):
                    sols[i] = DCSolution(sim.st, u[k], conv[k], ops[k])
            finally:
                sim.close()
        return SweepResult(pts, sols)
    mc = MNACircuit(target.circuit, target.params, MNASpec(temp=target.spec.temp, mode="dcop"))
    sim = BatchSimulator(mc, None, device)
    try:
        u, conv, _ = sim.dc(u0=u0)
        return DCSolution(sim.st, u[0], conv[0], sim.operating_points(u)[0])
    finally:
        sim.close()


class ACSol:
    """ac!'s result (src/ac.jl:75-82): the system linearised at the DC point -- G (with gmin on the voltage-node diagonals), C, the
    excitation b_ac -- plus the DC solution and the frequency grid in hertz.  ``sol[name]`` is the complex response of a node
    voltage or a branch current over the grid (ac.jl name-based access); ``freqresp(name, omegas)`` evaluates at angular
    frequencies (ac.jl:185-215); ``magnitude_db`` / ``phase_deg`` as in ac.jl:228-273."""

    def __init__(self, st, G, C, b_ac, dc_x, freqs):
        self.st, self.G, self.C, self.b_ac, self.dc_x, self.freqs = st, G, C, b_ac, dc_x, np.asarray(freqs, dtype=float)
        self._cache = {}

    def _solve(self, omegas):
        key = tuple(np.asarray(omegas, dtype=float))
        if key not in self._cache:
            self._cache[key] = np.array([np.linalg.solve(self.G + 1j * w * self.C, self.b_ac) for w in key]) if len(key) else np.zeros((0, self.st.n), complex)
        return self._cache[key]

    def freqresp(self, name, omegas):
        return self._solve(omegas)[:, self.st.index_of(name)]

    def __getitem__(self, name):
        return self.freqresp(name, 2.0 * np.pi * self.freqs)

    def magnitude_db(self, name, freqs=None):
        r = self[name] if freqs is None else self.freqresp(name, 2.0 * np.pi * np.asarray(freqs, dtype=float))
        return 20.0 * np.log10(np.abs(r))

    def phase_deg(self, name, freqs=None):
        r = self[name] if freqs is None else self.freqresp(name, 2.0 * np.pi * np.asarray(freqs, dtype=float))
        return np.degrees(np.angle(r))


def acdec(points_per_decade, fstart, fstop):
    """SPICE ``.ac dec``: logarithmic grid with ``points_per_decade`` points per decade from fstart to fstop (hertz)."""
    n = int(np.floor(np.log10(fstop / fstart) * points_per_decade + 1e-9)) + 1
    return fstart * 10.0 ** (np.arange(n) / points_per_decade)


def rhs_ac(st, circuit, params):
    """get_rhs_ac (build.jl:169-190): V sources stamp their ``ac`` value on their branch row (devices.jl:659), I sources +ac into p
    and -ac into n (devices.jl:728-729)."""
    from .circuit import resolve
    b = np.zeros(st.n, dtype=complex)
    for d in circuit.devices:
        if d.type not in ("V", "I"):
            continue
        ac = complex(resolve(d.params.get("ac", 0.0), params))
        if ac == 0:
            continue
        if d.type == "V":
            b[st.index_of("I_" + d.name)] += ac
        else:
            for nm, sgn in ((d.nodes[0], 1.0), (d.nodes[1], -1.0)):
                if nm not in ("0", "gnd", "gnd!"):
                    b[st.index_of(nm)] += sgn * ac
    return b


def ac(target, freqs=(), gmin=1e-12, device=0):
    """ac!(circuit, freqs; gmin) -- src/ac.jl:113-170.  The DC operating point and the restamp at it run on the GPU (cadnip_dc_run,
    cadnip_rebuild: the linearisation IS the stamping); G gets ``gmin`` on the voltage-node diagonals (assemble_G(ctx; gshunt=gmin),
    ac.jl:127).  The frequency sweep is the reference's own dense ``(jw C + G)^-1 b_ac`` on the host: n is a circuit's size, not a
    batch dimension.  A CircuitSweep returns one ACSol per point (one resident batch per structure class)."""
    import scipy.sparse as sp
    sweep = isinstance(target, CircuitSweep)
    mc0 = target.circuit if sweep else target
    mc = MNACircuit(mc0.circuit, mc0.params, MNASpec(temp=mc0.spec.temp, mode="dcop", gmin=mc0.spec.gmin))
    pts = target.points() if sweep else [{}]
    sols = [None] * len(pts)
    for idx, st in structure_classes(mc, pts) if sweep else [(list(range(1)), None)]:
        sim = BatchSimulator(mc, [pts[i] for i in idx] if sweep else None, device, st=st)
        try:
            st = sim.st
            u, conv, _ = sim.dc()
            if not np.all(conv):
                raise RuntimeError("ac: the DC operating point did not converge for %d point(s)" % int((~conv).sum()))
            sim.h.rebuild(u, 0.0)
            G, C, _, _ = sim.h.get_GCb()
            for k, i in enumerate(idx):
                dense = lambda nz: sp.csc_matrix((nz, st.ref_rowval, st.ref_colptr), shape=(st.n, st.n)).toarray()
                Gd, Cd = dense(G[k]), dense(C[k])
                Gd[np.arange(st.n_nodes), np.arange(st.n_nodes)] += gmin
                p_i = {kk: float(v[k]) for kk, v in sim.params.items()}
                sols[i] = ACSol(st, Gd, Cd, rhs_ac(st, mc.circuit, p_i), u[k].copy(), freqs)
        finally:
            sim.close()
    return SweepResult(pts, sols) if sweep else sols[0]


# ---- noise! (src/noise.jl) ---------------------------------------------------------------------------------------------------------------
K_BOLTZMANN, Q_ELEMENTARY = 1.380649e-23, 1.602176634e-19


class NoiseSol:
    """noise!'s result (src/noise.jl:31-40): output-referred noise PSD over a grid in hertz, per-source contributions (device names in lower
    case), and -- when an input source was named -- the input -> output gain and the input-referred PSD.  ``ns["onoise"]``, ``ns["inoise"]``,
    ``ns[source]`` as in noise.jl:239-253."""

    def __init__(self, freqs, output, onoise, contributions, temp, input, gain, inoise):
        self.freqs, self.output, self.onoise, self.contributions = freqs, output, onoise, contributions
        self.temp, self.input, self.gain, self.inoise = temp, input, gain, inoise

    def __getitem__(self, name):
        if name == "onoise":
            return self.onoise
        if name == "inoise":
            if self.input is None:
                raise KeyError("NoiseSol: no input-referred noise -- call noise(circuit, output, freqs, input=...)")
            return self.inoise
        if name in self.contributions:
            return self.contributions[name]
        raise KeyError("NoiseSol: unknown key %r (have onoise, inoise and the sources %s)" % (name, sorted(self.contributions)))


def total_noise(ns, referred="output"):
    """band-integrated RMS noise sqrt(int S df) by the trapezoidal rule (noise.jl:258-276)"""
    if referred not in ("output", "input"):
        raise ValueError("total_noise: referred must be 'output' or 'input', got %r" % (referred,))
    psd = ns["onoise"] if referred == "output" else ns["inoise"]
    fs = ns.freqs
    if len(fs) < 2:
        return float(np.sqrt(psd[0] if len(fs) == 1 else 0.0))
    return float(np.sqrt(np.sum(0.5 * (psd[:-1] + psd[1:]) * np.diff(fs))))


def noise_psd(src, temp_c, f):
    """one-sided PSD of a registered source (context.jl:179-189): thermal 4kT a, shot 2q a, white a, flicker a / f^b"""
    kind, a, b = src[2], src[3], src[4]
    if kind == "thermal":
        return 4.0 * K_BOLTZMANN * (float(temp_c) + 273.15) * a
    if kind == "shot":
        return 2.0 * Q_ELEMENTARY * a
    if kind == "white":
        return a
    return a / float(f) ** b


def noise_sources(st, circuit, params, u, temp_c=27.0, gmin=1e-12):
    """The noise sources of ``circuit`` at the solution ``u`` (one instance, [n]): (p, n, kind, a, b, name) with global unknown indices
    (-1 = ground).  What the reference's devices register while the builder runs at the DC point (context.jl:1017-1127): resistors their
    thermal noise 4kT/R (devices.jl:498-503), diodes the shot noise of their junction current and, with KF > 0, its flicker noise
    (devices.jl:1393-1443, 1582-1585), SimpleMOSFETs their channel thermal and flicker noise (devices.jl:1718-1732), instances of
    Verilog-A modules one source per white_noise / flicker_noise call of the contributions they execute (vasim.jl:2856-2893; evaluated by
    va/host_eval.py at the node voltages of ``u``: needs the model source)."""
    from .circuit import resolve
    from .opinfo import _index
    from . import va
    num = lambda v: float(np.asarray(resolve(v, params)).flat[0])
    out = []
    info = {d["name"]: d for d in st.opinfo}
    for d in circuit.devices:
        gl = [_index(st, t) for t in info[d.name]["nodes"]]
        name = d.name.lower()
        if d.type == "R":
            out.append((gl[0], gl[1], "thermal", 1.0 / num(d.params["r"]), 0.0, name))
        elif d.type in ("D", "DCAP"):
            v = (u[gl[0]] if gl[0] >= 0 else 0.0) - (u[gl[1]] if gl[1] >= 0 else 0.0)
            nVt = num(d.params["n"]) * num(d.params["Vt"])
            xarg = v / nVt
            i0 = num(d.params["Is"]) * ((np.exp(80.0) * (1.0 + (xarg - 80.0)) - 1.0) if xarg > 80.0 else (np.exp(xarg) - 1.0))
            out.append((gl[0], gl[1], "shot", abs(i0), 0.0, name))
            kf = num(d.params.get("KF", 0.0))
            if kf > 0:                           # devices.jl:1435-1443: KF |I0|^AF / f^FFE, under the device's own name like its shot noise
                out.append((gl[0], gl[1], "flicker", kf * abs(i0) ** num(d.params.get("AF", 1.0)), num(d.params.get("FFE", 1.0)), name))
        elif d.type.startswith("VA:"):
            mod = va.get(d.type[3:])[1]
            given = {k: num(v) for k, v in d.model.items()}
            par = va.host_eval.defaults(mod, given)
            V = [u[g] if g >= 0 else 0.0 for g in gl[:mod.n_nodes]]
            vold = [(V[p] if p >= 0 else 0.0) - (V[n] if n >= 0 else 0.0) for p, n in mod.limit_branches]   # at the solution the limit unknowns sit on their probes

            def on_noise(a, b, fn, pwr, expo, label, gl=gl, name=name):
                nm = ("%s_%s" % (name, label.lower())) if label else name
                out.append((gl[a] if a >= 0 else -1, gl[b] if b >= 0 else -1, "white" if fn == "white_noise" else "flicker", pwr, expo, nm))
            va.host_eval.evaluate(mod, V, par, temp_c + 273.15, num(d.params.get("m", 1.0)), gmin, vold=vold, given=set(d.model), mode="dcop", on_noise=on_noise)
        elif d.type == "MOS1":
            # the hand-written sp_mos1 device: its sources are those of the model text it transcribes (models/VADistillerModels.jl/va/mos1.va:
            # rd / rs thermal, channel thermal, flicker), evaluated by the host evaluator on that text at the solution
            mod = va.get("sp_mos1")[1]
            given = {k: num(v) for k, v in d.model.items()}
            par = va.host_eval.defaults(mod, given)
            V = [u[g] if g >= 0 else 0.0 for g in gl[:mod.n_nodes]]           # d, g, s, b, d_int, s_int: the module's node order
            vold = [(V[p] if p >= 0 else 0.0) - (V[n] if n >= 0 else 0.0) for p, n in mod.limit_branches]

            def on_noise(a, b, fn, pwr, expo, label, gl=gl, name=name):
                nm = ("%s_%s" % (name, label.lower())) if label else name
                out.append((gl[a] if a >= 0 else -1, gl[b] if b >= 0 else -1, "white" if fn == "white_noise" else "flicker", pwr, expo, nm))
            va.host_eval.evaluate(mod, V, par, temp_c + 273.15, num(d.params.get("m", 1.0)), gmin, vold=vold, given=set(d.model), mode="dcop", on_noise=on_noise)
        elif d.type == "SMOS":
            # SimpleMOSFET (devices.jl:1667-1732): channel thermal noise 4kT (2/3) gm between drain and source where the device conducts,
            # flicker noise KF |Ids|^AF / f^FFE when KF > 0 -- gm and Ids of the square law at the operating point
            vat = lambda k: u[gl[k]] if gl[k] >= 0 else 0.0
            vgs, vds = vat(1) - vat(2), vat(0) - vat(2)
            vth, kk, lam = num(d.params["Vth"]), num(d.params["K"]), num(d.params["lambda"])
            if vgs <= vth:
                ids = gm = 0.0
            elif vds <= vgs - vth:
                ids, gm = kk * ((vgs - vth) * vds - vds ** 2 / 2), kk * vds
            else:
                ids, gm = kk / 2 * (vgs - vth) ** 2 * (1 + lam * vds), kk * (vgs - vth) * (1 + lam * vds)
            if gm > 0:
                out.append((gl[0], gl[2], "thermal", 2.0 / 3.0 * gm, 0.0, name))
            kf = num(d.params.get("KF", 0.0))
            if kf > 0 and ids != 0.0:
                out.append((gl[0], gl[2], "flicker", kf * abs(ids) ** num(d.params.get("AF", 1.0)), num(d.params.get("FFE", 1.0)), name))
    return out


def noise_solve(st, G, C, sources, output, freqs, input=None, temp_c=27.0):
    """noise.jl:150-188 on dense G (gmin already on the voltage-node diagonals) and C: one adjoint solve per frequency,
    S_out(f) = sum_k |x_adj[p_k] - x_adj[n_k]|^2 S_k(f); the same adjoint gives the gain from the input source's branch row."""
    freqs = np.asarray(freqs, dtype=float)
    if freqs.size == 0:
        raise ValueError("noise(circuit, output, freqs=...) needs a non-empty grid in hertz (e.g. acdec(20, 1, 1e6))")
    names = list(st.node_names) + list(st.current_names)
    if output in ("gnd", "0"):
        raise KeyError("noise: the output cannot be ground")
    if output not in names:
        raise KeyError("noise: unknown output %s (nodes %s, currents %s)" % (output, st.node_names, st.current_names))
    out_idx = names.index(output)
    in_idx = None
    if input is not None:
        cand = [nm for nm in ("I_" + input, "I_" + input.lower()) if nm in st.current_names]
        if not cand:
            raise KeyError("noise: input source %s is not an independent voltage source (no current variable I_%s)" % (input, input))
        in_idx = st.n_nodes + st.current_names.index(cand[0])
    e_out = np.zeros(st.n, dtype=complex)
    e_out[out_idx] = 1.0
    onoise = np.zeros(len(freqs))
    contributions = {s[5]: np.zeros(len(freqs)) for s in sources}
    gain = np.zeros(len(freqs), dtype=complex) if input is not None else np.zeros(0, dtype=complex)
    inoise = np.zeros(len(freqs)) if input is not None else np.zeros(0)
    for fi, f in enumerate(freqs):
        x_adj = np.linalg.solve((1j * 2.0 * np.pi * f * C + G).T, e_out)
        for s in sources:
            Hk = (x_adj[s[0]] if s[0] >= 0 else 0.0) - (x_adj[s[1]] if s[1] >= 0 else 0.0)
            c = abs(Hk) ** 2 * noise_psd(s, temp_c, f)
            onoise[fi] += c
            contributions[s[5]][fi] += c
        if input is not None:
            H = x_adj[in_idx]
            gain[fi] = H
            inoise[fi] = np.inf if H == 0 else onoise[fi] / abs(H) ** 2
    return NoiseSol(freqs, output, onoise, contributions, temp_c, input, gain, inoise)


def noise(target, output, freqs, input=None, gmin=1e-12, device=0):
    """noise!(circuit, output; freqs, input, gmin) -- src/noise.jl:118-190.  As for ``ac``: the DC operating point and the restamp at it run on the
    GPU; the sources are collected on the host at that point (noise_sources) and the adjoint sweep is the reference's own dense solve."""
    import scipy.sparse as sp
    if len(freqs) == 0:
        raise ValueError("noise(circuit, output, freqs=...) needs a non-empty grid in hertz (e.g. acdec(20, 1, 1e6))")
    mc = MNACircuit(target.circuit, target.params, MNASpec(temp=target.spec.temp, mode="dcop", gmin=target.spec.gmin))
    sim = BatchSimulator(mc, None, device)
    try:
        st = sim.st
        u, conv, _ = sim.dc()
        if not np.all(conv):
            raise RuntimeError("noise: the DC operating point did not converge")
        sim.h.rebuild(u, 0.0)
        G, C, _, _ = sim.h.get_GCb()
        dense = lambda nz: sp.csc_matrix((nz, st.ref_rowval, st.ref_colptr), shape=(st.n, st.n)).toarray()
        Gd, Cd = dense(G[0]), dense(C[0])
        Gd[np.arange(st.n_nodes), np.arange(st.n_nodes)] += gmin
        p0 = {kk: float(v[0]) for kk, v in sim.params.items()}
        srcs = noise_sources(st, mc.circuit, p0, u[0], mc.spec.temp, mc.spec.gmin)
        return noise_solve(st, Gd, Cd, srcs, output, freqs, input, mc.spec.temp)
    finally:
        sim.close()


def tran(target, tspan, abstol=1e-10, reltol=1e-8, saveat=None, device=0, **kw):
    """tran!(circuit, tspan) / tran!(cs::CircuitSweep, tspan) -- sweeps.jl:588-665, 692-707."""
    tspan = (float(tspan[0]), float(tspan[1]))
    if saveat is None:
        saveat = np.linspace(tspan[0], tspan[1], 501)
    saveat = np.asarray(saveat, dtype=float)
    if isinstance(target, CircuitSweep):
        sim = BatchSimulator(target.circuit, target.points(), device)
    else:
        sim = BatchSimulator(target, None, device)
    try:
        out, per, stats = sim.tran(tspan, _resolve_abstol(abstol, sim.st), reltol, saveat, **kw)
        sols = []
        for i in range(sim.B):
            s = {"nnonliniter": int(per[i, 0]), "naccept": int(per[i, 1]), "nreject": int(per[i, 2])}
            sols.append(TranSolution(sim.st, saveat, out[i], s, "Success" if per[i, 3] == 1 else "Failure"))
        if isinstance(target, CircuitSweep):
            res = SweepResult(target.points(), sols)
            res.stats = stats
            return res
        sols[0].run_stats = stats
        return sols[0]
    finally:
        sim.close()
