"""ctypes binding of libcadnip_hip.so (include/cadnip_hip.h).

The library is the product; this module only marshals numpy arrays across the C ABI.  There
is no CPU fallback: if the extension is missing or no GPU is present, the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from .structure import Structure, TYPE_ID, type_id

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CADNIP_HIP_LIB") or os.path.join(_HERE, "libcadnip_hip.so")   # same override as julia/CadnipHIP.jl

OK, BADARG, SINGULAR, NONFINITE, HIPERROR, NOTREADY, NOCONV = range(7)
STATUS_NAMES = {0: "CADNIP_OK", 1: "CADNIP_BADARG", 2: "CADNIP_SINGULAR", 3: "CADNIP_NONFINITE",
                4: "CADNIP_HIPERROR", 5: "CADNIP_NOTREADY", 6: "CADNIP_NOCONV"}

# every symbol declared in include/cadnip_hip.h
EXPORTS = [
    "cadnip_create", "cadnip_destroy", "cadnip_set_params", "cadnip_set_spec", "cadnip_set_initjct",
    "cadnip_rebuild", "cadnip_residual", "cadnip_jacobian", "cadnip_jacobian_dense", "cadnip_ode_rhs", "cadnip_ode_jacobian", "cadnip_get_GCb", "cadnip_get_contributions", "cadnip_analyze",
    "cadnip_analyze_values", "cadnip_factor", "cadnip_solve", "cadnip_newton_step", "cadnip_newton_step_fused", "cadnip_debug_step_time", "cadnip_lu_stats", "cadnip_dc_run",
    "cadnip_dc_log_size", "cadnip_dc_log_get", "cadnip_tran_run", "cadnip_tran_state", "cadnip_dev_ptr", "cadnip_stream", "cadnip_set_u", "cadnip_get_u", "cadnip_get_flags",
    "cadnip_sync", "cadnip_debug_copy", "cadnip_debug_stamp_time", "cadnip_profile_enable", "cadnip_profile_read", "cadnip_version",
    "cadnip_host_lu_analyze", "cadnip_host_lu_analyze_leaves", "cadnip_host_lu_size", "cadnip_host_lu_blocks", "cadnip_host_lu_get", "cadnip_host_lu_free",
    "cadnip_host_f2_build", "cadnip_host_f2_size", "cadnip_host_f2_get", "cadnip_host_f2_free", "cadnip_host_f2_team_steps", "cadnip_host_f2_steps",
]


class CadnipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s returned %s" % (where, STATUS_NAMES.get(code, code)))


class SingularException(CadnipError):
    """Maps CADNIP_SINGULAR like the Julia shim maps it to LinearAlgebra.SingularException."""


class DeviceBlockC(C.Structure):
    _fields_ = [("type", C.c_int32), ("count", C.c_int32),
                ("n_nodes", C.c_int32), ("nodes", C.POINTER(C.c_int32)),
                ("n_ipar", C.c_int32), ("ipar", C.POINTER(C.c_int32)),
                ("n_par", C.c_int32),
                ("g_base", C.c_int32), ("c_base", C.c_int32), ("b_base", C.c_int32),
                ("n_g", C.c_int32), ("n_c", C.c_int32), ("n_b", C.c_int32)]


_I = C.POINTER(C.c_int32)
_D = C.POINTER(C.c_double)


class StructureC(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_nodes", C.c_int32), ("n_currents", C.c_int32), ("n_charges", C.c_int32),
                ("n_limits", C.c_int32), ("nnz", C.c_int32), ("rowptr", _I), ("colidx", _I), ("to_ref_nz", _I),
                ("n_blocks", C.c_int32), ("blocks", C.POINTER(DeviceBlockC)),
                ("n_wave_data", C.c_int32), ("wave_data", _D),
                ("ns_g", C.c_int32), ("ns_c", C.c_int32), ("ns_b", C.c_int32),
                ("g_ptr", _I), ("g_slots", _I), ("c_ptr", _I), ("c_slots", _I), ("b_ptr", _I), ("b_slots", _I),
                ("diag_nz", _I), ("limit_init", _D)]


class SpecC(C.Structure):
    _fields_ = [("mode", C.c_int32), ("gmin", C.c_double), ("gshunt", C.c_double), ("srcFact", C.c_double)]


class DCOptsC(C.Structure):
    _fields_ = [("abstol", C.c_double), ("maxiters", C.c_int32), ("use_pcnr", C.c_int32), ("cold_start", C.c_int32),
                ("use_stepping", C.c_int32), ("fused", C.c_int32), ("participate", _I)]


class TranOptsC(C.Structure):
    _fields_ = [("t0", C.c_double), ("t1", C.c_double), ("reltol", C.c_double), ("abstol", _D), ("err_mask", _D), ("h0", C.c_double),
                ("hmin", C.c_double), ("hmax", C.c_double), ("max_newton", C.c_int32), ("max_order", C.c_int32),
                ("use_pcnr", C.c_int32), ("newton_tol", C.c_double), ("n_break", C.c_int32), ("breaks", _D),
                ("n_save", C.c_int32), ("save_t", _D), ("n_obs", C.c_int32), ("obs", _I),
                ("max_iterations", C.c_int64), ("fused", C.c_int32), ("newton_mode", C.c_int32), ("step_rule", C.c_int32)]


class RunStatsC(C.Structure):
    _fields_ = [("newton_iters", C.c_int64), ("steps_accepted", C.c_int64), ("steps_rejected", C.c_int64),
                ("newton_failures", C.c_int64), ("launches", C.c_int64), ("n_failed", C.c_int32),
                ("wall_seconds", C.c_double)]


_lib = None


def _share_hip_runtime():
    """One HIP runtime per process.  The library links /opt/rocm's libamdhip64.so.7; a PyTorch wheel carries its own copy and
    asks for it by file name (libamdhip64.so), which the dynamic loader does not match against a copy that was loaded under
    its soname -- so a torch imported AFTER this library brought a second HIP / HSA runtime into the process and found no
    device (round 1 had to order its tests around that).  If torch is installed but not loaded yet, its copy is loaded
    first, globally, under the name torch will ask for: the library (which asks for the soname) and a later torch then
    resolve to the same runtime, exactly as when torch comes first.  CADNIP_HIP_RUNTIME=<path> names another runtime to
    share, CADNIP_HIP_RUNTIME=system skips this."""
    import importlib.util
    import sys
    want = os.environ.get("CADNIP_HIP_RUNTIME", "")
    if want == "system" or "torch" in sys.modules:
        return
    cand = want
    if not cand:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen the in-tree extension; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    _share_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.cadnip_version.restype = C.c_char_p
    lib.cadnip_dev_ptr.restype = C.c_void_p
    lib.cadnip_stream.restype = C.c_void_p
    lib.cadnip_destroy.restype = None
    lib.cadnip_host_lu_free.restype = None
    lib.cadnip_host_lu_size.restype = C.c_int32
    lib.cadnip_dev_ptr.argtypes = [C.c_void_p, C.c_int32]
    lib.cadnip_stream.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def _ip(a):
    return a.ctypes.data_as(_I)


def _dp(a):
    return a.ctypes.data_as(_D)


def _check(code, where):
    if code == OK:
        return
    if code == SINGULAR:
        raise SingularException(code, where)
    raise CadnipError(code, where)


MODE = {"dcop": 0, "tran": 1, "tranop": 2}

MODE_NAMES = ("dcop", "tran", "tranop")


class Handle:
    """One (structure, GPU) handle with ``B`` resident sweep instances."""

    def __init__(self, st: Structure, B: int = 1, device: int = 0):
        self.lib = load_library()
        self.st = st
        self.B = int(B)
        self._keep = []
        blocks = (DeviceBlockC * max(1, len(st.blocks)))()
        for k, blk in enumerate(st.blocks):
            nodes = np.ascontiguousarray(blk.nodes, dtype=np.int32)
            ipar = np.ascontiguousarray(blk.ipar, dtype=np.int32)
            self._keep += [nodes, ipar]
            b = blocks[k]
            b.type, b.count = type_id(blk.type), blk.count
            b.n_nodes, b.nodes = nodes.shape[0], _ip(nodes)
            b.n_ipar, b.ipar = ipar.shape[0], _ip(ipar)
            b.n_par = blk.n_par
            b.g_base, b.c_base, b.b_base = blk.g_base, blk.c_base, blk.b_base
            b.n_g, b.n_c, b.n_b = blk.n_g, blk.n_c, blk.n_b
        s = StructureC()
        s.n, s.n_nodes, s.n_currents, s.n_charges, s.n_limits = st.n, st.n_nodes, st.n_currents, st.n_charges, st.n_limits
        s.nnz = st.nnz
        arrs = {}
        for nm in ("rowptr", "colidx", "to_ref_nz", "g_ptr", "g_slots", "c_ptr", "c_slots", "b_ptr", "b_slots", "diag_nz"):
            a = np.ascontiguousarray(getattr(st, nm), dtype=np.int32)
            if a.size == 0:
                a = np.zeros(1, dtype=np.int32)
            arrs[nm] = a
            setattr(s, nm, _ip(a))
        wd = np.ascontiguousarray(st.wave_data, dtype=np.float64)
        wdp = wd if wd.size else np.zeros(1)
        li = np.ascontiguousarray(st.limit_init, dtype=np.float64)
        lip = li if li.size else np.zeros(1)
        self._keep += [arrs, wdp, lip, blocks]
        s.n_blocks, s.blocks = len(st.blocks), blocks
        s.n_wave_data, s.wave_data = wd.size, _dp(wdp)
        s.ns_g, s.ns_c, s.ns_b = st.ns_g, st.ns_c, st.ns_b
        s.limit_init = _dp(lip)
        self.h = C.c_void_p()
        _check(self.lib.cadnip_create(C.byref(s), C.c_int32(self.B), C.c_int32(device), C.byref(self.h)), "cadnip_create")
        self.spec = dict(mode="tran", gmin=1e-12, gshunt=0.0, srcFact=1.0)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.cadnip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters / spec --------------------------------------------------------------
    def set_params(self, packed):
        for k, arr in enumerate(packed):
            a = np.ascontiguousarray(arr, dtype=np.float64)
            want = (self.B, self.st.blocks[k].n_par, self.st.blocks[k].count)
            if a.shape != want:
                raise ValueError("parameter block %d has shape %r, the structure needs %r" % (k, a.shape, want))
            _check(self.lib.cadnip_set_params(self.h, C.c_int32(k), _dp(a)), "cadnip_set_params")

    def set_spec(self, mode=None, gmin=None, gshunt=None, srcFact=None):
        for k, v in (("mode", mode), ("gmin", gmin), ("gshunt", gshunt), ("srcFact", srcFact)):
            if v is not None:
                self.spec[k] = v
        sp = SpecC(MODE[self.spec["mode"]], self.spec["gmin"], self.spec["gshunt"], self.spec["srcFact"])
        _check(self.lib.cadnip_set_spec(self.h, C.byref(sp)), "cadnip_set_spec")

    def set_initjct(self, on):
        _check(self.lib.cadnip_set_initjct(self.h, C.c_int32(1 if on else 0)), "cadnip_set_initjct")

    # -- the three callbacks --------------------------------------------------------------
    def _bn(self, x):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (self.B, self.st.n)))
        return a

    def _b(self, x):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (self.B,)))

    def rebuild(self, u, t=0.0):
        """fast_rebuild!(ws, u, t)"""
        ua, ta = self._bn(u), self._b(t)
        _check(self.lib.cadnip_rebuild(self.h, _dp(ua), _dp(ta)), "cadnip_rebuild")

    def residual(self, du, u):
        """fast_residual! minus the restamp: resid = C*du + G*u - b at the last rebuild."""
        dua, ua = self._bn(du), self._bn(u)
        r = np.empty((self.B, self.st.n))
        _check(self.lib.cadnip_residual(self.h, _dp(dua), _dp(ua), _dp(r)), "cadnip_residual")
        return r

    def jacobian(self, gamma, readback=True):
        """fast_jacobian! minus the restamp: J = G + gamma*C; returned in the reference's CSC nz order."""
        g = self._b(gamma)
        J = np.empty((self.B, self.st.nnz)) if readback else None
        _check(self.lib.cadnip_jacobian(self.h, _dp(g), _dp(J) if readback else None), "cadnip_jacobian")
        return J

    def jacobian_dense(self, gamma):
        """fast_jacobian!(J::Matrix, ...) of a dense structure (precompile.jl:588-603) minus the restamp: [B, n, n] with J[b, i, j]."""
        g = self._b(gamma)
        J = np.empty((self.B, self.st.n * self.st.n))
        _check(self.lib.cadnip_jacobian_dense(self.h, _dp(g), _dp(J)), "cadnip_jacobian_dense")
        return J.reshape(self.B, self.st.n, self.st.n).transpose(0, 2, 1)        # column-major blocks -> [i, j]

    def ode_rhs(self, u, t=0.0):
        """rhs!(du, u, p, t) of the ODE form (src/mna/solve.jl:2241-2248): restamp, du = b - G*u."""
        ua, ta = self._bn(u), self._b(t)
        du = np.empty((self.B, self.st.n))
        _check(self.lib.cadnip_ode_rhs(self.h, _dp(ua), _dp(ta), _dp(du)), "cadnip_ode_rhs")
        return du

    def ode_jacobian(self, u, t=0.0):
        """jac!(J, u, p, t) of the ODE form (src/mna/solve.jl:2251-2276): restamp, J = -G in the reference's nz order."""
        ua, ta = self._bn(u), self._b(t)
        J = np.empty((self.B, self.st.nnz))
        _check(self.lib.cadnip_ode_jacobian(self.h, _dp(ua), _dp(ta), _dp(J)), "cadnip_ode_jacobian")
        return J

    def get_GCb(self):
        B, st = self.B, self.st
        G, Cm, b = np.empty((B, st.nnz)), np.empty((B, st.nnz)), np.empty((B, st.n))
        lw = np.empty((B, max(st.n_limits, 1)))
        _check(self.lib.cadnip_get_GCb(self.h, _dp(G), _dp(Cm), _dp(b), _dp(lw)), "cadnip_get_GCb")
        return G, Cm, b, lw[:, :st.n_limits]

    def get_contributions(self):
        """Per-device contributions of one restamp at the current state: (S_g [B, ns_g], S_c [B, ns_c], S_b [B, ns_b])."""
        st = self.st
        ns = st.ns_g + st.ns_c + st.ns_b
        S = np.zeros((self.B, max(ns, 1)))
        _check(self.lib.cadnip_get_contributions(self.h, _dp(S)), "cadnip_get_contributions")
        return S[:, :st.ns_g], S[:, st.ns_g:st.ns_g + st.ns_c], S[:, st.ns_g + st.ns_c:ns]

    # -- LU ----------------------------------------------------------------------------------
    def analyze(self, sample_instance=0):
        _check(self.lib.cadnip_analyze(self.h, C.c_int32(sample_instance)), "cadnip_analyze")

    def analyze_values(self, J_ref_nz):
        """``J_ref_nz``: sample values in the reference's CSC nz order."""
        csr = np.ascontiguousarray(np.asarray(J_ref_nz, dtype=np.float64)[self.st.to_ref_nz])
        _check(self.lib.cadnip_analyze_values(self.h, _dp(csr)), "cadnip_analyze_values")

    def factor(self):
        _check(self.lib.cadnip_factor(self.h), "cadnip_factor")

    def solve(self, rhs):
        r = self._bn(rhs)
        x = np.empty_like(r)
        _check(self.lib.cadnip_solve(self.h, _dp(r), _dp(x)), "cadnip_solve")
        return x

    def newton_step(self, u, du, gamma=None, t=None, refresh=True, want_resid=False, fused=False):
        """cadnip_newton_step: resid = C du + G u - b, [J = G + gamma C refactored,] delta = J^-1 resid -- one call, one synchronisation.
        ``fused``: cadnip_newton_step_fused (one kernel; agrees to rounding).  Returns (delta [B, n], ||resid||_2 [B]) and resid [B, n]
        with ``want_resid``."""
        uu, dd = self._bn(u), self._bn(du)
        delta, nrm = np.empty_like(uu), np.empty(self.B)
        resid = np.empty_like(uu) if want_resid else None
        g = None if gamma is None else np.ascontiguousarray(np.broadcast_to(np.asarray(gamma, dtype=np.float64), (self.B,)))
        tt = None if t is None else np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        fn = self.lib.cadnip_newton_step_fused if fused else self.lib.cadnip_newton_step
        _check(fn(self.h, _dp(uu), _dp(dd), None if g is None else _dp(g), None if tt is None else _dp(tt), C.c_int32(1 if refresh else 0),
                  _dp(delta), _dp(nrm), None if resid is None else _dp(resid)), "cadnip_newton_step_fused" if fused else "cadnip_newton_step")
        return (delta, nrm, resid) if want_resid else (delta, nrm)

    def lu_stats(self):
        v = [C.c_int32() for _ in range(5)]
        _check(self.lib.cadnip_lu_stats(self.h, *[C.byref(x) for x in v]), "cadnip_lu_stats")
        return dict(zip(("nnz_lu", "n_terms", "n_levels", "n_fwd_levels", "n_bwd_levels"), [x.value for x in v]))

    # -- drivers -------------------------------------------------------------------------------
    def dc_run(self, u0=None, abstol=1e-10, maxiters=100, use_pcnr=True, cold_start=True, use_stepping=True,
               raise_on_fail=False, fused=False, participate=None):
        """``participate``: boolean mask [B]; instances with False sit the run out (u unchanged, converged False)."""
        u = self._bn(0.0 if u0 is None else u0).copy()
        conv = np.zeros(self.B, dtype=np.int32)
        st = RunStatsC()
        pm = None if participate is None else np.ascontiguousarray(np.asarray(participate, dtype=bool).astype(np.int32))
        o = DCOptsC(abstol, maxiters, int(use_pcnr), int(cold_start), int(use_stepping), int(fused), None if pm is None else _ip(pm))
        rc = self.lib.cadnip_dc_run(self.h, C.byref(o), _dp(u), _ip(conv), C.byref(st))
        if rc not in (OK, NOCONV) or (rc == NOCONV and raise_on_fail):
            _check(rc, "cadnip_dc_run")
        return u, conv.astype(bool), _stats(st)

    def dc_log(self):
        """The fallback chain of the last ``dc_run``: list of (instance, stage, rung value, converged, Newton solves)."""
        k = int(self.lib.cadnip_dc_log_size(self.h))
        inst, stage, ok = (np.zeros(max(k, 1), dtype=np.int32) for _ in range(3))
        val, it = np.zeros(max(k, 1)), np.zeros(max(k, 1), dtype=np.int64)
        _check(self.lib.cadnip_dc_log_get(self.h, _ip(inst), _ip(stage), _dp(val), _ip(ok), it.ctypes.data_as(C.POINTER(C.c_int64))), "cadnip_dc_log_get")
        return [(int(inst[j]), int(stage[j]), float(val[j]), bool(ok[j]), int(it[j])) for j in range(k)]

    def tran_run(self, t0, t1, abstol, reltol=1e-4, breaks=(), save_t=(), obs=None, h0=0.0, hmin=0.0, hmax=0.0,
                 max_newton=10, max_order=2, use_pcnr=False, newton_tol=1e-3, max_iterations=0, fused=False,
                 err_mask="differential", newton_mode=0, step_rule=0):
        at = np.ascontiguousarray(np.broadcast_to(np.asarray(abstol, dtype=np.float64), (self.st.n,)))
        if isinstance(err_mask, str):
            em = self.st.differential_mask() if err_mask == "differential" else np.ones(self.st.n)
        else:
            em = np.ones(self.st.n) if err_mask is None else np.asarray(err_mask, dtype=np.float64)
        em = np.ascontiguousarray(em, dtype=np.float64)
        br = np.ascontiguousarray(np.asarray(breaks, dtype=np.float64))
        sv = np.ascontiguousarray(np.asarray(save_t, dtype=np.float64))
        ob = np.ascontiguousarray(np.asarray(obs if obs is not None else [], dtype=np.int32))
        n_obs = ob.size if ob.size else self.st.n
        out = np.zeros((self.B, sv.size, n_obs))
        per = np.zeros((self.B, 4), dtype=np.int64)
        st = RunStatsC()
        o = TranOptsC(t0, t1, reltol, _dp(at), _dp(em), h0, hmin, hmax, max_newton, max_order, int(use_pcnr), newton_tol,
                      br.size, _dp(br) if br.size else None, sv.size, _dp(sv) if sv.size else None,
                      ob.size, _ip(ob) if ob.size else None, max_iterations, int(fused), int(newton_mode), int(step_rule))  # fused: 0 = one kernel per op, non-zero = fused Newton kernel
        rc = self.lib.cadnip_tran_run(self.h, C.byref(o), _dp(out), per.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(st))
        if rc not in (OK, NOCONV):
            _check(rc, "cadnip_tran_run")
        return out, per, _stats(st)

    def tran_state(self):
        t, hh, o = np.empty(self.B), np.empty(self.B), np.empty(self.B, dtype=np.int32)
        _check(self.lib.cadnip_tran_state(self.h, _dp(t), _dp(hh), _ip(o)), "cadnip_tran_state")
        return t, hh, o

    # -- misc ------------------------------------------------------------------------------------
    def set_u(self, u):
        _check(self.lib.cadnip_set_u(self.h, _dp(self._bn(u))), "cadnip_set_u")

    def get_u(self):
        u = np.empty((self.B, self.st.n))
        _check(self.lib.cadnip_get_u(self.h, _dp(u)), "cadnip_get_u")
        return u

    def debug_copy(self, n_doubles, reps=1):
        _check(self.lib.cadnip_debug_copy(self.h, C.c_int64(n_doubles), C.c_int32(reps)), "cadnip_debug_copy")

    def stamp_time(self, block=-1, reps=20):
        """Milliseconds of ``reps`` back-to-back launches of one block's stamping kernel (block < 0: the whole restamp)."""
        ms = C.c_double()
        _check(self.lib.cadnip_debug_stamp_time(self.h, C.c_int32(block), C.c_int32(reps), C.byref(ms)), "cadnip_debug_stamp_time")
        return ms.value

    def profile(self, on=True):
        _check(self.lib.cadnip_profile_enable(self.h, C.c_int32(1 if on else 0)), "cadnip_profile_enable")

    def profile_read(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_double * 64)()
        calls = (C.c_int64 * 64)()
        k = self.lib.cadnip_profile_read(self.h, C.c_int32(64), names, ms, calls)
        return {names[i].decode(): (ms[i], calls[i]) for i in range(k)}


def _stats(st):
    return {f: getattr(st, f) for f, _ in RunStatsC._fields_}


LU_ARRAYS = ("rperm", "cperm", "rowptr", "col", "diag", "load_src", "load_dst", "ent_pos", "ent_diag", "ent_ptr",
             "term_a", "term_b", "lev_ptr", "fwd_rows", "fwd_lev_ptr", "bwd_rows", "bwd_lev_ptr")


F2_ARRAYS = (("posW", np.int32), ("lanes", np.uint64), ("passes", np.uint64), ("terms", np.uint32), ("meta", np.int32))


def leaves_of(st):
    """(q_begin, lim_begin, unit_ok) of a Structure: the leaf-first pivot order a handle built from it uses (csrc/api.hip: cadnip_create)."""
    q0, l0 = st.n_nodes + st.n_currents, st.n - st.n_limits
    ok = np.zeros(st.n, dtype=np.uint8)
    rows = np.repeat(np.arange(st.n), np.diff(st.rowptr))
    for e in np.flatnonzero((rows == np.asarray(st.colidx)) & (rows >= q0)):
        ok[rows[e]] = 1 if (st.g_ptr[e + 1] - st.g_ptr[e] == 1 and st.c_ptr[e + 1] == st.c_ptr[e]) else 0
    return q0, l0, ok


def host_lu_analyze(n, rowptr, colidx, vals, pivot_tol=1e-3, sample=False, f2_nc=None, leaves=None, order=None):
    """Host-only symbolic phase (no GPU needed): returns the LU program as a dict of int32 arrays.  ``leaves``: ``leaves_of(st)`` for
    the pivot order of a handle (device-local unknowns first), None for the plain Markowitz search.  ``order``: "klu" | "markowitz" forces
    the ordering (csrc/symbolic.cpp: default Markowitz up to 4 096 unknowns, KLU's block triangular form + minimum degree beyond);
    ``out["n_blocks"]`` = diagonal blocks found (0 with the Markowitz search)."""
    if order is not None:
        prev = os.environ.get("CADNIP_LU_ORDER")
        os.environ["CADNIP_LU_ORDER"] = order
        try:
            return host_lu_analyze(n, rowptr, colidx, vals, pivot_tol, sample, f2_nc, leaves)
        finally:
            if prev is None:
                os.environ.pop("CADNIP_LU_ORDER", None)
            else:
                os.environ["CADNIP_LU_ORDER"] = prev
    lib = load_library()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colidx, dtype=np.int32)
    v = np.ascontiguousarray(vals, dtype=np.float64)
    p = C.c_void_p()
    if leaves is None or os.environ.get("CADNIP_LU_NOLEAF"):
        _check(lib.cadnip_host_lu_analyze(C.c_int32(n), _ip(rp), _ip(ci), _dp(v), C.c_double(pivot_tol), C.c_int32(int(sample)), C.byref(p)),
               "cadnip_host_lu_analyze")
    else:
        ok = np.ascontiguousarray(leaves[2], dtype=np.uint8)
        _check(lib.cadnip_host_lu_analyze_leaves(C.c_int32(n), _ip(rp), _ip(ci), _dp(v), C.c_double(pivot_tol), C.c_int32(int(sample)),
                                                 C.c_int32(int(leaves[0])), C.c_int32(int(leaves[1])), ok.ctypes.data_as(C.c_void_p), C.byref(p)),
               "cadnip_host_lu_analyze_leaves")
    out = {}
    try:
        for k, nm in enumerate(LU_ARRAYS):
            sz = lib.cadnip_host_lu_size(p, C.c_int32(k))
            a = np.zeros(max(sz, 1), dtype=np.int32)
            _check(lib.cadnip_host_lu_get(p, C.c_int32(k), _ip(a)), "cadnip_host_lu_get")
            out[nm] = a[:sz]
        lib.cadnip_host_lu_blocks.restype = C.c_int32
        out["n_blocks"] = int(lib.cadnip_host_lu_blocks(p))
        if f2_nc is not None:
            ts = (C.c_int32 * 4)()
            out["team_steps"] = {}
            for nw in (1, 2, 4):
                if lib.cadnip_host_f2_team_steps(p, C.c_int32(int(f2_nc)), C.c_int32(nw), ts) == 0:
                    out["team_steps"][nw] = tuple(int(v) for v in ts)
            # ... as straight-line steps: "steps" = {nw: (counts, uint64 descriptor words)}; nw = 1 is the sweep kernel's three-term layout
            out["steps"] = {}
            for nw in (1, 2, 4, 12, 14):
                if lib.cadnip_host_f2_steps(p, C.c_int32(int(f2_nc)), C.c_int32(nw), ts, None) == 0:
                    words = np.zeros(max(int(ts[3]), 1), dtype=np.uint64)
                    _check(lib.cadnip_host_f2_steps(p, C.c_int32(int(f2_nc)), C.c_int32(nw), ts, words.ctypes.data_as(C.c_void_p)), "cadnip_host_f2_steps")
                    out["steps"][nw] = (tuple(int(v) for v in ts[:3]), words[:int(ts[3])])
            # the fused kernel's entry program for this LU and core size (csrc/f2_program.cpp)
            lib.cadnip_host_f2_free.restype = None
            q = C.c_void_p()
            _check(lib.cadnip_host_f2_build(p, C.c_int32(int(f2_nc)), C.byref(q)), "cadnip_host_f2_build")
            try:
                f2 = {}
                for k, (nm, dt) in enumerate(F2_ARRAYS):
                    sz = lib.cadnip_host_f2_size(q, C.c_int32(k))
                    a = np.zeros(max(sz, 1), dtype=dt)
                    _check(lib.cadnip_host_f2_get(q, C.c_int32(k), a.ctypes.data_as(C.c_void_p)), "cadnip_host_f2_get")
                    f2[nm] = a[:sz]
                out["f2"] = f2
            finally:
                lib.cadnip_host_f2_free(q)
    finally:
        lib.cadnip_host_lu_free(p)
    return out
