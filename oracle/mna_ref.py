"""TEST INFRASTRUCTURE (oracle) -- literal CPU restatement of Cadnip.jl's MNA hot path.

Pure Python + numpy/scipy, scalar loops, small cases only.  Each function cites
the reference file:line it follows (paths relative to /root/reference).

Covered: MNAContext structure discovery (src/mna/context.jl), assembly
(src/mna/build.jl), compile_structure / EvalWorkspace / fast_rebuild! /
fast_residual! / fast_jacobian! (src/mna/precompile.jl), DirectStampContext
positional stamping (src/mna/value_only.jl), builtin devices and waves
(src/mna/devices.jl), PCNR Newton + fallback chain (src/mna/solve.jl:542-929).
KLU is replaced by scipy.sparse.linalg.splu (SuperLU); KLU itself (SuiteSparse,
unpinned version, via LinearSolve.jl compat "2, 3, 4.2, 5.0") is not vendored
in the reference.
"""
import math
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .dual import val

CHARGE_SCALE = 1e12  # contrib.jl:39

# typed indices (context.jl:47-110): plain ints are node indices; tuples tag the rest
GROUND = 0


def CurrentIndex(k):
    return ("c", k)


def ChargeIndex(k):
    return ("q", k)


def LimitIndex(k):
    return ("l", k)


def _iszero(i):
    return isinstance(i, int) and i == 0


class ZeroVector:
    """context.jl:120-125: returns 0.0 for any index, length 0."""

    def __len__(self):
        return 0

    def __getitem__(self, i):
        return 0.0


ZERO_VECTOR = ZeroVector()


class MNASpec:
    """solve.jl:57-70."""

    def __init__(self, temp=27.0, mode="tran", time=0.0, gmin=1e-12, gshunt=0.0, srcFact=1.0,
                 tnom=27.0, abstol=1e-12, reltol=1e-3, vntol=1e-6, iabstol=1e-12):
        self.temp = float(temp)
        self.mode = mode
        self.time = time
        self.gmin = gmin
        self.gshunt = gshunt
        self.srcFact = srcFact
        self.tnom = tnom
        self.abstol = abstol
        self.reltol = reltol
        self.vntol = vntol
        self.iabstol = iabstol

    def replace(self, **kw):
        d = dict(self.__dict__)
        d.update(kw)
        return MNASpec(**d)


def x_at(x, i):
    """1-based read of the solution vector, tolerant of a short x (vasim.jl:3123-3133)."""
    if i == 0:
        return 0.0
    if i <= len(x):
        return float(x[i - 1])
    return 0.0


# =============================================================================
# MNAContext (structure discovery)                         context.jl:248-372
# =============================================================================
class MNAContext:
    direct = False

    def __init__(self):
        self.node_names = []
        self.node_to_idx = {}
        self.n_nodes = 0
        self.internal_node_flags = []
        self.current_names = []
        self.n_currents = 0
        self.G_I, self.G_J, self.G_V = [], [], []
        self.C_I, self.C_J, self.C_V = [], [], []
        self.b_I, self.b_V = [], []
        self.charge_names = []
        self.n_charges = 0
        self.charge_branches = []
        self.charge_is_vdep = []
        self.charge_Q_values = []
        self.charge_V_values = []
        self.charge_detection_pos = 0
        self.limit_names = []
        self.n_limits = 0
        self.limit_branches = []
        self.limit_init = []
        self.limit_w = []
        self.breakpoints = []
        self.initjct = False
        self.noise = []          # deferred noise-source channel (context.jl:281-292): (p, n, kind, a, b, name)

    # -- noise-source channel (context.jl:1017-1127): a small-signal noise current between p and n with PSD described by (kind, a, b)
    def stamp_noise(self, p, n, kind, a, b, name):
        self.noise.append((p, n, kind, float(a), float(b), str(name).lower()))

    def register_thermal_noise(self, p, n, G, name):      # S = 4 k T G
        self.stamp_noise(p, n, "thermal", G, 0.0, name)

    def register_channel_thermal_noise(self, p, n, gm, name, gamma=2.0 / 3.0):   # context.jl:1076-1077: the THERMAL shape with the conductance gamma gm
        self.stamp_noise(p, n, "thermal", gamma * gm, 0.0, name)

    def register_shot_noise(self, p, n, I, name):         # S = 2 q |I|
        self.stamp_noise(p, n, "shot", abs(I), 0.0, name)

    def register_white_noise(self, p, n, pwr, name):      # Verilog-A white_noise(pwr): S = pwr
        self.stamp_noise(p, n, "white", pwr, 0.0, name)

    def register_flicker_noise(self, p, n, pwr, expo, name):   # Verilog-A flicker_noise(pwr, exp): S = pwr / f^exp
        self.stamp_noise(p, n, "flicker", pwr, expo, name)

    # -- allocation -------------------------------------------------------------
    def get_node(self, name):  # context.jl:467-490
        if isinstance(name, int):
            return name
        if name in ("gnd", "0", "gnd!"):
            return 0
        i = self.node_to_idx.get(name)
        if i is None:
            self.n_nodes += 1
            i = self.n_nodes
            self.node_names.append(name)
            self.node_to_idx[name] = i
            self.internal_node_flags.append(False)
        return i

    def alloc_internal_node(self, name):  # context.jl:654-686
        i = self.node_to_idx.get(name)
        if i is None:
            i = self.get_node(name)
            self.internal_node_flags[i - 1] = True
        return i

    def alloc_current(self, name):  # context.jl:523-560
        self.n_currents += 1
        self.current_names.append(name)
        return CurrentIndex(self.n_currents)

    def get_current_idx(self, name):  # context.jl:591
        return CurrentIndex(self.current_names.index(name) + 1)

    def alloc_charge(self, name, p, n):  # context.jl:741-768
        if name in self.charge_names:
            return ChargeIndex(self.charge_names.index(name) + 1)
        self.n_charges += 1
        self.charge_names.append(name)
        self.charge_branches.append((p, n))
        return ChargeIndex(self.n_charges)

    def alloc_limit(self, name, p, n, init=0.0):  # context.jl:826-857
        if name in self.limit_names:
            return LimitIndex(self.limit_names.index(name) + 1)
        self.n_limits += 1
        self.limit_names.append(name)
        self.limit_branches.append((p, n))
        self.limit_init.append(float(init))
        self.limit_w.append(float(init))
        return LimitIndex(self.n_limits)

    def record_limit_w(self, lidx, w):  # context.jl:859
        self.limit_w[lidx[1] - 1] = float(w)

    def system_size(self):  # context.jl:438
        return self.n_nodes + self.n_currents + self.n_charges + self.n_limits

    def resolve_index(self, idx):  # context.jl:577-581
        if isinstance(idx, int):
            return idx
        kind, k = idx
        if kind == "c":
            return self.n_nodes + k
        if kind == "q":
            return self.n_nodes + self.n_currents + k
        return self.n_nodes + self.n_currents + self.n_charges + k

    # -- stamping -------------------------------------------------------------
    def stamp_G(self, i, j, v):  # context.jl:945-953
        if _iszero(i) or _iszero(j):
            return
        self.G_I.append(i)
        self.G_J.append(j)
        self.G_V.append(val(v))

    def stamp_C(self, i, j, v):  # context.jl:969-977
        if _iszero(i) or _iszero(j):
            return
        self.C_I.append(i)
        self.C_J.append(j)
        self.C_V.append(val(v))

    def stamp_b(self, i, v):  # context.jl:994-999
        if _iszero(i):
            return
        self.b_I.append(i)
        self.b_V.append(val(v))

    def detect_or_cached(self, name, V_branch, Q):  # contrib.jl:214-257
        pos = self.charge_detection_pos
        self.charge_detection_pos = pos + 1
        V = float(V_branch)
        Qv = float(Q)
        if pos >= len(self.charge_Q_values):
            self.charge_is_vdep.append(False)
            self.charge_Q_values.append(Qv)
            self.charge_V_values.append(V)
            return False
        Vs = self.charge_V_values[pos]
        Qs = self.charge_Q_values[pos]
        V_min = 1e-6
        if abs(V) > V_min and abs(Vs) > V_min:
            Cc = Qv / V
            Cs = Qs / Vs
            diff = abs(Cc - Cs)
            maxC = max(abs(Cc), abs(Cs))
            if diff > 1e-15 and (maxC < 1e-30 or diff / maxC > 1e-6):
                self.charge_is_vdep[pos] = True
        self.charge_Q_values[pos] = Qv
        self.charge_V_values[pos] = V
        return self.charge_is_vdep[pos]

    def reset_detection_counter(self):  # contrib.jl:330
        self.charge_detection_pos = 0

    def register_breakpoints(self, wave):  # context.jl:1606
        if wave is not None and hasattr(wave, "breakpoints"):
            bp = wave.breakpoints()
            if bp is not None:
                self.breakpoints.append(bp)

    def reset_for_restamping(self):  # context.jl:1528-1600 (detection cache preserved)
        keep = (self.charge_is_vdep, self.charge_Q_values, self.charge_V_values)
        initjct = self.initjct
        self.__init__()
        self.charge_is_vdep, self.charge_Q_values, self.charge_V_values = keep
        self.initjct = initjct


def stamp_conductance(ctx, p, n, G):  # context.jl:1362-1368
    ctx.stamp_G(p, p, G)
    ctx.stamp_G(p, n, -G)
    ctx.stamp_G(n, p, -G)
    ctx.stamp_G(n, n, G)


def stamp_capacitance(ctx, p, n, C):  # context.jl:1376-1382
    ctx.stamp_C(p, p, C)
    ctx.stamp_C(p, n, -C)
    ctx.stamp_C(n, p, -C)
    ctx.stamp_C(n, n, C)


# =============================================================================
# Assembly                                                      build.jl:81-230
# =============================================================================
def _resolved(ctx, idxs):
    return np.array([ctx.resolve_index(i) for i in idxs], dtype=np.int64)


def assemble_G(ctx):
    n = ctx.system_size()
    I = _resolved(ctx, ctx.G_I)
    J = _resolved(ctx, ctx.G_J)
    return sp.coo_matrix((np.array(ctx.G_V, dtype=float), (I - 1, J - 1)), shape=(n, n)).tocsc()


def assemble_C(ctx):
    n = ctx.system_size()
    I = _resolved(ctx, ctx.C_I)
    J = _resolved(ctx, ctx.C_J)
    return sp.coo_matrix((np.array(ctx.C_V, dtype=float), (I - 1, J - 1)), shape=(n, n)).tocsc()


def get_rhs(ctx):
    b = np.zeros(ctx.system_size())
    for i, v in zip(ctx.b_I, ctx.b_V):
        k = ctx.resolve_index(i)
        if k > 0:
            b[k - 1] += v
    return b


class MNAData:  # build.jl:39-51
    def __init__(self, ctx):
        self.G = assemble_G(ctx)
        self.C = assemble_C(ctx)
        self.b = get_rhs(ctx)
        self.node_names = list(ctx.node_names)
        self.current_names = list(ctx.current_names)
        self.charge_names = list(ctx.charge_names)
        self.limit_names = list(ctx.limit_names)
        self.n_nodes = ctx.n_nodes
        self.n_currents = ctx.n_currents
        self.n_charges = ctx.n_charges
        self.n_limits = ctx.n_limits

    def index_of(self, name):  # SII lookup order nodes -> currents -> charges -> limits (build.jl:421-457)
        if name in self.node_names:
            return self.node_names.index(name) + 1
        if name in self.current_names:
            return self.n_nodes + self.current_names.index(name) + 1
        if name in self.charge_names:
            return self.n_nodes + self.n_currents + self.charge_names.index(name) + 1
        if name in self.limit_names:
            return self.n_nodes + self.n_currents + self.n_charges + self.limit_names.index(name) + 1
        raise KeyError(name)


def assemble(ctx):
    return MNAData(ctx)


def state_abstol(sys, vntol=1e-6, iabstol=1e-12, chgtol=1e-14):  # build.jl:276-283
    n = sys.n_nodes + sys.n_currents + sys.n_charges + sys.n_limits
    tol = np.empty(n)
    a = sys.n_nodes
    b = a + sys.n_currents
    c = b + sys.n_charges
    tol[:a] = vntol
    tol[a:b] = iabstol
    tol[b:c] = chgtol
    tol[c:] = vntol
    return tol


def detect_differential_vars(sys):  # solve.jl:2041-2058
    n = sys.G.shape[0]
    C = sys.C.tocoo()
    d = np.zeros(n, dtype=bool)
    for i, v in zip(C.row, C.data):
        if abs(v) > 1e-30:
            d[i] = True
    return d


# =============================================================================
# DirectStampContext                                     value_only.jl:42-478
# =============================================================================
class DirectStampContext:
    direct = True

    def __init__(self, ctx, G_nzval, C_nzval, b, G_mapping, C_mapping, b_resolved):
        self.node_to_idx = ctx.node_to_idx
        self.n_nodes = ctx.n_nodes
        self.n_currents = ctx.n_currents
        self.n_charges = ctx.n_charges
        self.current_names = ctx.current_names
        self.G_nzval = G_nzval
        self.C_nzval = C_nzval
        self.G_mapping = G_mapping
        self.C_mapping = C_mapping
        self.b = b
        self.b_V = np.zeros(len(ctx.b_V))
        self.b_resolved = b_resolved
        self.charge_is_vdep = list(ctx.charge_is_vdep)
        self.limit_w = np.array(ctx.limit_w, dtype=float)
        self.internal_node_indices = [i + 1 for i, f in enumerate(ctx.internal_node_flags) if f]
        self.initjct = False
        self.overflow = False
        self._reset_counters()

    def _reset_counters(self):
        self.G_pos = 0
        self.C_pos = 0
        self.b_pos = 0
        self.current_pos = 0
        self.charge_pos = 0
        self.limit_pos = 0
        self.charge_detection_pos = 0
        self.internal_node_pos = 0

    def reset(self):  # reset_direct_stamp!  value_only.jl:238-261
        self._reset_counters()
        self.G_nzval[:] = 0.0
        self.C_nzval[:] = 0.0
        self.b[:] = 0.0
        self.b_V[:] = 0.0

    def get_node(self, name):  # value_only.jl:267-273
        if isinstance(name, int):
            return name
        if name in ("gnd", "0", "gnd!"):
            return 0
        return self.node_to_idx[name]

    def alloc_internal_node(self, name):  # value_only.jl:275-290
        i = self.internal_node_indices[self.internal_node_pos]
        self.internal_node_pos += 1
        return i

    def alloc_current(self, name):  # value_only.jl:294-343
        self.current_pos += 1
        return CurrentIndex(self.current_pos)

    def get_current_idx(self, name):  # value_only.jl:207
        return CurrentIndex(self.current_names.index(name) + 1)

    def alloc_charge(self, name, p, n):  # value_only.jl:302-355
        self.charge_pos += 1
        return ChargeIndex(self.charge_pos)

    def alloc_limit(self, name, p, n, init=0.0):  # value_only.jl:357-377
        self.limit_pos += 1
        return LimitIndex(self.limit_pos)

    def record_limit_w(self, lidx, w):  # value_only.jl:384
        self.limit_w[lidx[1] - 1] = float(w)

    def resolve_index(self, idx):  # value_only.jl:221-231
        if isinstance(idx, int):
            return idx
        kind, k = idx
        if kind == "c":
            return self.n_nodes + k
        if kind == "q":
            return self.n_nodes + self.n_currents + k
        return self.n_nodes + self.n_currents + self.n_charges + k

    def stamp_G(self, i, j, v):  # value_only.jl:395-421
        if _iszero(i) or _iszero(j):
            return
        pos = self.G_pos
        self.G_pos = pos + 1
        if pos >= len(self.G_mapping):
            self.overflow = True
            return
        nz = self.G_mapping[pos]
        if nz > 0:
            self.G_nzval[nz - 1] += val(v)

    def stamp_C(self, i, j, v):  # value_only.jl:428-451
        if _iszero(i) or _iszero(j):
            return
        pos = self.C_pos
        self.C_pos = pos + 1
        if pos >= len(self.C_mapping):
            self.overflow = True
            return
        nz = self.C_mapping[pos]
        if nz > 0:
            self.C_nzval[nz - 1] += val(v)

    def stamp_b(self, i, v):  # value_only.jl:459-478
        if _iszero(i):
            return
        pos = self.b_pos
        self.b_pos = pos + 1
        if pos >= len(self.b_V):
            self.overflow = True
            return
        self.b_V[pos] = val(v)

    def detect_or_cached(self, name, V_branch, Q):  # contrib.jl:279-283
        pos = self.charge_detection_pos
        self.charge_detection_pos = pos + 1
        return self.charge_is_vdep[pos]

    def reset_detection_counter(self):
        self.charge_detection_pos = 0

    def register_breakpoints(self, wave):  # value_only.jl:149
        return None


# =============================================================================
# compile_structure / EvalWorkspace                       precompile.jl:253-467
# =============================================================================
def compute_coo_to_nz_mapping(I, J, S):  # precompile.jl:253-283 (1-based nz indices; 0 = not mapped)
    S = S.tocsc()
    mapping = np.zeros(len(I), dtype=np.int64)
    for k in range(len(I)):
        i, j = int(I[k]), int(J[k])
        if i == 0 or j == 0:
            continue
        for idx in range(S.indptr[j - 1], S.indptr[j]):
            if S.indices[idx] == i - 1:
                mapping[k] = idx + 1
                break
        if mapping[k] == 0:
            raise RuntimeError("COO entry (%d,%d) not found" % (i, j))
    return mapping


def _pattern(I, J, n):
    """sparse(I,J,ones) pattern in CSC with sorted rows (precompile.jl:414-417)."""
    M = sp.coo_matrix((np.ones(len(I)), (np.asarray(I) - 1, np.asarray(J) - 1)), shape=(n, n)).tocsc()
    M.sum_duplicates()
    M.sort_indices()
    return M


class CompiledStructure:  # precompile.jl:75-124, 312-443
    def __init__(self, builder, params, spec, ctx=None):
        self.builder = builder
        self.params = params
        self.spec = spec
        ctx0 = ctx if ctx is not None else builder(params, spec, 0.0, x=ZERO_VECTOR)
        self.ctx0 = ctx0
        n = ctx0.system_size()
        self.n = n
        self.n_nodes = ctx0.n_nodes
        self.n_currents = ctx0.n_currents
        self.n_charges = ctx0.n_charges
        self.node_names = list(ctx0.node_names)
        self.current_names = list(ctx0.current_names)
        GI, GJ = _resolved(ctx0, ctx0.G_I), _resolved(ctx0, ctx0.G_J)
        CI, CJ = _resolved(ctx0, ctx0.C_I), _resolved(ctx0, ctx0.C_J)
        self.G_I, self.G_J, self.C_I, self.C_J = GI, GJ, CI, CJ
        self.G_n_coo, self.C_n_coo = len(GI), len(CI)
        self.n_b_deferred = len(ctx0.b_I)
        self.b_deferred_resolved = _resolved(ctx0, ctx0.b_I)
        pat = _pattern(np.concatenate([GI, CI]), np.concatenate([GJ, CJ]), n)
        self.colptr = pat.indptr.copy()
        self.rowval = pat.indices.copy()
        nnz = pat.nnz
        # padded G, C share colptr/rowval (precompile.jl:419-421)
        self.G = sp.csc_matrix((np.zeros(nnz), self.rowval, self.colptr), shape=(n, n))
        self.C = sp.csc_matrix((np.zeros(nnz), self.rowval, self.colptr), shape=(n, n))
        self.G_coo_to_idx = compute_coo_to_nz_mapping(GI, GJ, pat)
        self.C_coo_to_idx = compute_coo_to_nz_mapping(CI, CJ, pat)
        for k, v in enumerate(ctx0.G_V):
            self.G.data[self.G_coo_to_idx[k] - 1] += v
        for k, v in enumerate(ctx0.C_V):
            self.C.data[self.C_coo_to_idx[k] - 1] += v
        # diagonal nz indices for gshunt (precompile.jl:451-467)
        self.G_diag_idx = np.zeros(self.n_nodes, dtype=np.int64)
        for col in range(self.n_nodes):
            for idx in range(self.colptr[col], self.colptr[col + 1]):
                if self.rowval[idx] == col:
                    self.G_diag_idx[col] = idx + 1
                    break
        self.n_limits = ctx0.n_limits
        self.limit_init = np.array(ctx0.limit_init, dtype=float)

    def with_spec(self, spec):
        import copy
        c = copy.copy(self)
        c.spec = spec
        return c


def compile_structure(builder, params, spec, ctx=None):
    return CompiledStructure(builder, params, spec, ctx=ctx)


class EvalWorkspace:  # precompile.jl:168-222
    def __init__(self, cs, ctx=None):
        self.structure = cs
        ctx = ctx if ctx is not None else cs.ctx0
        self.b = np.zeros(cs.n)
        self.dctx = DirectStampContext(ctx, cs.G.data, cs.C.data, self.b, cs.G_coo_to_idx,
                                       cs.C_coo_to_idx, cs.b_deferred_resolved)
        self.resid_tmp = np.zeros(cs.n)


def create_workspace(cs, ctx=None):
    return EvalWorkspace(cs, ctx)


def fast_rebuild(ws, u, t, cs=None):  # precompile.jl:493-537
    cs = cs if cs is not None else ws.structure
    d = ws.dctx
    d.reset()
    cs.builder(cs.params, cs.spec, float(t), x=u, ctx=d)
    for k in range(cs.n_b_deferred):
        idx = d.b_resolved[k]
        if idx > 0:
            d.b[idx - 1] += d.b_V[k]
    if cs.spec.srcFact < 1.0:
        d.b *= cs.spec.srcFact
    g = cs.spec.gshunt
    if g != 0.0:
        for i in range(cs.n_nodes):
            idx = cs.G_diag_idx[i]
            if idx > 0:
                d.G_nzval[idx - 1] += g


def fast_residual(resid, du, u, ws, t):  # precompile.jl:546-557
    fast_rebuild(ws, u, t)
    cs = ws.structure
    resid[:] = cs.C @ np.asarray(du) + cs.G @ np.asarray(u) - ws.dctx.b


def fast_jacobian(J_nz, du, u, ws, gamma, t):  # precompile.jl:568-585
    fast_rebuild(ws, u, t)
    cs = ws.structure
    J_nz[:] = cs.G.data + gamma * cs.C.data


def ode_rhs(du, u, ws, t):  # rhs! of the ODE form, solve.jl:2241-2248: du = b - G*u
    fast_rebuild(ws, u, t)
    du[:] = ws.dctx.b - ws.structure.G @ np.asarray(u)


def ode_jac(J_nz, u, ws, t):  # jac! of the ODE form, solve.jl:2251-2276: J = -G (mass matrix cs.C constant)
    fast_rebuild(ws, u, t)
    J_nz[:] = -ws.structure.G.data


# =============================================================================
# Structure detection                                      solve.jl:1793-1822
# =============================================================================
def build_with_detection(builder, params, spec, seed=0xDEADBEEF):
    """Five builder passes; random x in [-1,1].  The reference seeds
    MersenneTwister(0xDEADBEEF); numpy's stream differs, which only changes the
    probe points, not the (structural) detection outcome."""
    rng = np.random.default_rng(seed)
    ctx = None
    for _ in range(5):
        if ctx is None:
            ctx = builder(params, spec, 0.0, x=ZERO_VECTOR)
        else:
            known = ctx.system_size()
            ctx.reset_for_restamping()
            x = (rng.random(known) - 0.5) * 2.0
            builder(params, spec, 0.0, x=x, ctx=ctx)
    return ctx


# =============================================================================
# DC Newton                                                  solve.jl:542-929
# =============================================================================
def _lu_solve(cs, F):
    A = sp.csc_matrix((cs.G.data.copy(), cs.rowval, cs.colptr), shape=(cs.n, cs.n))
    try:
        with np.errstate(all="ignore"):
            lu = spla.splu(A)
            return lu.solve(F)
    except RuntimeError:
        return None


def dc_newton_plain(cs, ws, u0, abstol=1e-10, maxiters=100):
    """Stand-in for _dc_newton_compiled (solve.jl:542-578).  The reference hands the
    same residual/Jacobian to NonlinearSolve's polyalgorithm (third-party, unpinned);
    plain Newton-Raphson on F = G*u - b with the same ||F||_2 < abstol stop is the
    first member of that chain."""
    u = np.array(u0, dtype=float)
    iters = 0
    for it in range(maxiters + 1):
        fast_rebuild(ws, u, 0.0, cs)
        F = cs.G @ u - ws.dctx.b
        if not np.all(np.isfinite(F)):
            return u, False, iters
        if np.linalg.norm(F) < abstol:
            return u, True, iters
        if it == maxiters:
            break
        d = _lu_solve(cs, F)
        if d is None or not np.all(np.isfinite(d)):
            return u, False, iters
        u = u - d
        iters += 1
    return u, False, iters


def dc_pcnr_newton(cs, ws, u0, abstol=1e-10, maxiters=100):  # solve.jl:599-698
    n = len(u0)
    L = cs.n_limits
    if L == 0:
        return np.array(u0, dtype=float), False, 0
    lim0 = n - L
    u = np.array(u0, dtype=float)
    if not np.any(u):
        u[lim0:] = cs.limit_init
        ws.dctx.initjct = True
    try:
        for it in range(1, maxiters + 1):
            fast_rebuild(ws, u, 0.0, cs)
            ws.dctx.initjct = False
            F = cs.G @ u - ws.dctx.b
            if not np.all(np.isfinite(F)):
                return u, False, it - 1
            if np.linalg.norm(F) < abstol:
                u[lim0:] = ws.dctx.limit_w
                fast_rebuild(ws, u, 0.0, cs)
                F = cs.G @ u - ws.dctx.b
                if np.linalg.norm(F) < abstol:
                    return u, True, it - 1
            d = _lu_solve(cs, F)
            if d is None or not np.all(np.isfinite(d)):
                return u, False, it - 1
            u -= d
            u[lim0:] = ws.dctx.limit_w
        return u, False, maxiters
    finally:
        ws.dctx.initjct = False


def gshunt_stepping(cs, ws, u0, abstol=1e-10, maxiters=100, gshunt_start=1e-3, gshunt_factor=10.0,
                    max_steps=20):  # solve.jl:720-783
    target = cs.spec.gshunt
    g = gshunt_start
    gmin = max(target, 1e-12)
    u = np.array(u0, dtype=float)
    saved = u.copy()
    converged = False
    for _ in range(max_steps):
        cs_step = cs.with_spec(cs.spec.replace(gshunt=g))
        un, ok, _ = dc_newton_plain(cs_step, ws, u, abstol, maxiters)
        if ok:
            u = un.copy()
            saved = u.copy()
            if g <= gmin:
                if g != target:
                    cs_f = cs.with_spec(cs.spec.replace(gshunt=target))
                    uf, okf, _ = dc_newton_plain(cs_f, ws, u, abstol, maxiters)
                    if okf:
                        u = uf
                        converged = True
                else:
                    converged = True
                break
            g /= gshunt_factor
            if g < gmin:
                g = gmin
        else:
            if gshunt_factor <= 1.5:
                break
            gshunt_factor = math.sqrt(gshunt_factor)
            u = saved.copy()
    return u, converged


def source_stepping(cs, ws, u0, abstol=1e-10, maxiters=100, raise_=0.1, max_steps=50):  # solve.jl:805-850
    src = 0.0
    conv = 0.0
    u = np.array(u0, dtype=float)
    saved = u.copy()
    for _ in range(max_steps):
        cs_step = cs.with_spec(cs.spec.replace(srcFact=src))
        un, ok, _ = dc_newton_plain(cs_step, ws, u, abstol, maxiters)
        if ok:
            conv = src
            u = un.copy()
            saved = u.copy()
            if src >= 1.0:
                return u, True
            src = min(src + raise_, 1.0)
        else:
            if src - conv < 1e-6:
                break
            raise_ /= 2.0
            src = conv + raise_
            u = saved.copy()
    return u, False


def dc_solve_with_fallbacks(cs, ws, u0, abstol=1e-10, maxiters=100, use_stepping=True):  # solve.jl:871-929
    n = len(u0)
    if n == 0:
        return np.zeros(0), True
    if cs.n_limits > 0:
        try:
            u, ok, _ = dc_pcnr_newton(cs, ws, u0, abstol, maxiters)
        except (ValueError, OverflowError, ZeroDivisionError):
            u, ok = u0, False
        if ok:
            return u, True
    u, ok, _ = dc_newton_plain(cs, ws, u0, abstol, maxiters)
    if ok:
        return u, True
    if not use_stepping:
        return u, False
    u, ok = gshunt_stepping(cs, ws, np.zeros(n), abstol, maxiters)
    if ok:
        return u, True
    u, ok = source_stepping(cs, ws, np.zeros(n), abstol, maxiters)
    return u, ok


class DCSolution:  # solve.jl:156-166
    def __init__(self, sys, x, converged):
        self.sys = sys
        self.x = x
        self.converged = converged

    def __getitem__(self, name):
        return self.x[self.sys.index_of(name) - 1]


def solve_dc(builder, params, spec, abstol=1e-10, maxiters=100, u0=None):  # solve.jl:2389-2419, 973-988
    ctx = build_with_detection(builder, params, spec)
    n = ctx.system_size()
    cs = compile_structure(builder, params, spec, ctx=ctx)
    ws = create_workspace(cs, ctx=ctx)
    u_init = np.zeros(n) if u0 is None or len(u0) != n else np.array(u0, dtype=float)
    u, ok = dc_solve_with_fallbacks(cs, ws, u_init, abstol, maxiters)
    ctx.reset_for_restamping()
    builder(params, spec, 0.0, x=u, ctx=ctx)
    return DCSolution(assemble(ctx), u, ok)


def dc(builder, params=None, spec=None, u0=None):
    """dc!(circuit) == solve_dc(with_mode(circuit, :dcop))   sweeps.jl:450-455."""
    spec = spec if spec is not None else MNASpec()
    spec = MNASpec(temp=spec.temp, mode="dcop")  # with_mode keeps only temp+mode (solve.jl:1976-1989)
    return solve_dc(builder, params or {}, spec, u0=u0)


# ---- noise analysis (src/noise.jl:118-190, context.jl:173-189) ------------------------------------------------------------------
K_BOLTZMANN, Q_ELEMENTARY = 1.380649e-23, 1.602176634e-19


def noise_psd(src, temp_c, f):  # context.jl:179-189
    _, _, kind, a, b, _ = src
    if kind == "thermal":
        return 4 * K_BOLTZMANN * (float(temp_c) + 273.15) * a
    if kind == "shot":
        return 2 * Q_ELEMENTARY * a
    if kind == "white":
        return a
    return a / float(f) ** b


def noise(builder, params, spec, output, freqs, input=None, gmin=1e-12):
    """noise!(circuit, output; freqs, input, gmin) -- noise.jl:118-190: DC point, rebuild at it (the sources register themselves), one adjoint
    solve (jw C + G)^T x = e_out per frequency, S_out = sum_k |x_p - x_n|^2 S_k.  Returns (onoise, contributions by source name, gain, inoise)."""
    freqs = [float(f) for f in freqs]
    if not freqs:
        raise ValueError("noise!(circuit, output; freqs=...) needs a non-empty Hz grid")
    spec = MNASpec(temp=spec.temp, mode="dcop")
    sol = solve_dc(builder, params, spec)
    ctx = MNAContext()
    builder(params, spec, 0.0, x=ZERO_VECTOR, ctx=ctx)
    ctx.reset_for_restamping()
    ctx.noise = []
    builder(params, spec, 0.0, x=sol.x, ctx=ctx)
    sysm = assemble(ctx)
    n = ctx.system_size()
    G = np.array(sysm.G.toarray() if hasattr(sysm.G, "toarray") else sysm.G, dtype=float)
    C = np.array(sysm.C.toarray() if hasattr(sysm.C, "toarray") else sysm.C, dtype=float)
    for i in range(ctx.n_nodes):
        G[i, i] += gmin                                      # assemble_G(ctx; gshunt=gmin)
    names = list(ctx.node_names) + list(ctx.current_names)
    if output in ("gnd", "0") or output not in names:
        raise KeyError("noise!: unknown output %s" % output)
    out_idx = names.index(output)
    in_idx = None
    if input is not None:
        cands = [nm for nm in ("I_" + input, "I_" + input.lower()) if nm in ctx.current_names]
        if not cands:
            raise KeyError("noise!: input source %s is not an independent voltage source" % input)
        in_idx = ctx.n_nodes + ctx.current_names.index(cands[0])
    e_out = np.zeros(n, dtype=complex)
    e_out[out_idx] = 1.0
    onoise = np.zeros(len(freqs))
    contributions = {}
    for src in ctx.noise:
        contributions.setdefault(src[5], np.zeros(len(freqs)))
    gain = np.zeros(len(freqs), dtype=complex) if input is not None else None
    inoise = np.zeros(len(freqs)) if input is not None else None
    for fi, f in enumerate(freqs):
        w = 2 * np.pi * f
        x_adj = np.linalg.solve((1j * w * C + G).T, e_out)
        for src in ctx.noise:
            pp, qq = ctx.resolve_index(src[0]), ctx.resolve_index(src[1])
            Hk = (0.0 if pp == 0 else x_adj[pp - 1]) - (0.0 if qq == 0 else x_adj[qq - 1])
            c = abs(Hk) ** 2 * noise_psd(src, spec.temp, f)
            onoise[fi] += c
            contributions[src[5]][fi] += c
        if input is not None:
            H = x_adj[in_idx]
            gain[fi] = H
            inoise[fi] = np.inf if H == 0 else onoise[fi] / abs(H) ** 2
    return onoise, contributions, gain, inoise


# ---- AC small-signal response at a DC point (src/ac.jl:113-170, 185-215) -----------------------------------------------------
def ac_response(G, C, b_ac, omegas, n_nodes, gshunt=1e-12):
    """ac!: rebuild at the DC solution, G assembled with ``gshunt = gmin`` on the voltage-node diagonals
    (assemble_G(ctx; gshunt=gmin), ac.jl:127), descriptor system E = C, A = -G, B = b_ac, identity output;
    freqresp: x(jw) = (jw C + G)^-1 b_ac for every state.  ``G``, ``C`` dense [n, n]; returns [len(omegas), n] complex."""
    G = np.array(G, dtype=float)
    C = np.asarray(C, dtype=float)
    for i in range(n_nodes):
        G[i, i] += gshunt
    return np.array([np.linalg.solve(G + 1j * w * C, np.asarray(b_ac, dtype=complex)) for w in omegas])
