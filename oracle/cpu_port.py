"""TEST INFRASTRUCTURE (oracle) -- ctypes wrapper + build recipe of oracle/cpu_port.cpp.

The port consumes plain arrays: the structure arrays of one circuit (CSR pattern, device blocks,
slot gather lists), one sweep instance's parameter blocks, and an LU program (pivot sequence +
fill pattern).  The test harness / bench.py hand those over; nothing under oracle/ is imported
by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "cpu_port.cpp")
LIB = os.path.join(_HERE, "_build", "libcpu_port.so")

_I = C.POINTER(C.c_int32)
_D = C.POINTER(C.c_double)


def build(force=False):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    return LIB


class TranOptsC(C.Structure):
    _fields_ = [("t0", C.c_double), ("t1", C.c_double), ("reltol", C.c_double), ("abstol", _D), ("err_mask", _D),
                ("h0", C.c_double), ("hmin", C.c_double), ("hmax", C.c_double), ("max_newton", C.c_int32),
                ("max_order", C.c_int32), ("use_pcnr", C.c_int32), ("newton_tol", C.c_double),
                ("n_break", C.c_int32), ("breaks", _D), ("n_save", C.c_int32), ("save_t", _D),
                ("n_obs", C.c_int32), ("obs", _I), ("newton_mode", C.c_int32), ("step_rule", C.c_int32)]


class TranStatsC(C.Structure):
    _fields_ = [("newton_iters", C.c_int64), ("accepted", C.c_int64), ("rejected", C.c_int64),
                ("newton_failures", C.c_int64), ("status", C.c_int32), ("wall_seconds", C.c_double), ("refactorisations", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.port_create.restype = C.c_void_p
        _lib.port_destroy.restype = None
        _lib.port_add_block.restype = None
        _lib.port_set_spec.restype = None
        _lib.port_set_lu.restype = None
        _lib.port_rebuild.restype = None
    return _lib


def _ia(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a if a.size else np.zeros(1, dtype=np.int32)


def _da(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if a.size else np.zeros(1)


def _ip(a):
    return a.ctypes.data_as(_I)


def _dp(a):
    return a.ctypes.data_as(_D)


MODE = {"dcop": 0, "tran": 1, "tranop": 2}


class Port:
    """One circuit instance on the CPU.  ``st`` is any object with the structure arrays (the product's
    Structure dataclass is what the tests pass); ``packed`` the per-block parameter arrays [n_par, count]
    of ONE instance; ``type_id`` maps block type names to CadnipDeviceType ids."""

    def __init__(self, st, packed, type_id):
        L = lib()
        self.st = st
        self.n = st.n
        a = [_ia(x) for x in (st.rowptr, st.colidx, st.g_ptr, st.g_slots, st.c_ptr, st.c_slots, st.b_ptr, st.b_slots, st.diag_nz)]
        li, wv = _da(st.limit_init), _da(st.wave_data)
        self.p = C.c_void_p(L.port_create(st.n, st.n_nodes, st.n_limits, st.nnz, _ip(a[0]), _ip(a[1]), st.ns_g, st.ns_c, st.ns_b,
                                          _ip(a[2]), _ip(a[3]), _ip(a[4]), _ip(a[5]), _ip(a[6]), _ip(a[7]), _ip(a[8]),
                                          _dp(li), int(np.asarray(st.wave_data).size), _dp(wv)))
        for blk, par in zip(st.blocks, packed):
            nodes, ipar, pr = _ia(blk.nodes), _ia(blk.ipar), _da(par)
            assert pr.shape == (blk.n_par, blk.count) or blk.count == 0, pr.shape
            L.port_add_block(self.p, type_id[blk.type], blk.count, int(np.asarray(blk.nodes).shape[0]), _ip(nodes),
                             int(np.asarray(blk.ipar).shape[0]), _ip(ipar), blk.n_par, _dp(pr),
                             blk.g_base, blk.c_base, blk.b_base, blk.n_g, blk.n_c, blk.n_b)
        self.spec = dict(mode="tran", gmin=1e-12, gshunt=0.0, srcFact=1.0, initjct=0)
        self.set_spec()

    def close(self):
        if self.p:
            lib().port_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_spec(self, **kw):
        self.spec.update(kw)
        s = self.spec
        lib().port_set_spec(self.p, MODE[s["mode"]], C.c_double(s["gmin"]), C.c_double(s["gshunt"]), C.c_double(s["srcFact"]),
                            int(s["initjct"]))

    def set_stamper(self, fn):
        """External stamping: ``fn(u [n], t) -> (G_csr [nnz], C_csr [nnz], b [n], limit_w [n] or None)`` replaces the port's own device code in
        fast_rebuild! (circuits of generated Verilog-A models: the literal interpreter stamps, the port's controller and LU run)."""
        n, nnz = self.n, self.st.nnz
        proto = C.CFUNCTYPE(None, _D, C.c_double, _D, _D, _D, _D)

        def cb(u, t, G, Cm, b, lw):
            g, c, bb, l = fn(np.ctypeslib.as_array(u, (n,)).copy(), float(t))
            np.ctypeslib.as_array(G, (nnz,))[:] = g
            np.ctypeslib.as_array(Cm, (nnz,))[:] = c
            np.ctypeslib.as_array(b, (n,))[:] = bb
            if l is not None:
                np.ctypeslib.as_array(lw, (n,))[:] = l
        self._stamper = proto(cb)            # keep the trampoline alive
        lib().port_set_stamp_callback.restype = None
        lib().port_set_stamp_callback(self.p, self._stamper)

    def set_lu(self, prog):
        """``prog``: dict of int32 arrays (rperm cperm rowptr col diag load_src load_dst ent_pos ent_diag ent_ptr term_a term_b)."""
        k = {nm: _ia(prog[nm]) for nm in ("rperm", "cperm", "rowptr", "col", "diag", "load_src", "load_dst", "ent_pos", "ent_diag",
                                          "ent_ptr", "term_a", "term_b")}
        lib().port_set_lu(self.p, int(prog["rowptr"][-1]), _ip(k["rperm"]), _ip(k["cperm"]), _ip(k["rowptr"]), _ip(k["col"]),
                          _ip(k["diag"]), _ip(k["load_src"]), _ip(k["load_dst"]), int(len(prog["ent_pos"])), _ip(k["ent_pos"]),
                          _ip(k["ent_diag"]), _ip(k["ent_ptr"]), _ip(k["term_a"]), _ip(k["term_b"]))

    def rebuild(self, u, t=0.0):
        """fast_rebuild!: returns (G, C, b, limit_w) with G, C in the structure's CSR order."""
        st = self.st
        u = _da(u)
        G, Cm, b, lw = np.empty(st.nnz), np.empty(st.nnz), np.empty(st.n), np.empty(st.n)
        lib().port_rebuild(self.p, _dp(u), C.c_double(t), _dp(G), _dp(Cm), _dp(b), _dp(lw))
        return G, Cm, b, lw[st.n - st.n_limits:]

    def factor_solve(self, gamma, rhs):
        x = np.empty(self.n)
        rc = lib().port_factor_solve(self.p, C.c_double(gamma), _dp(_da(rhs)), _dp(x))
        if rc:
            raise RuntimeError("singular")
        return x

    def dc(self, u0=None, abstol=1e-10, maxiters=100, use_pcnr=True, cold_start=True):
        u = np.zeros(self.n) if u0 is None else np.array(u0, dtype=np.float64)
        it = C.c_int32()
        ok = lib().port_dc(self.p, _dp(u), C.c_double(abstol), maxiters, int(use_pcnr), int(cold_start), C.byref(it))
        return u, bool(ok), it.value

    def tran(self, u0, t0, t1, abstol, reltol, breaks=(), save_t=(), obs=None, err_mask=None, h0=0.0, hmin=0.0, hmax=0.0,
             max_newton=10, max_order=2, use_pcnr=True, newton_tol=1e-3, trace_cap=0, newton_mode=0, step_rule=0):
        n = self.n
        u = np.array(u0, dtype=np.float64)
        at = _da(np.broadcast_to(np.asarray(abstol, dtype=np.float64), (n,)))
        em = _da(np.ones(n) if err_mask is None else err_mask)
        br, sv = _da(breaks), _da(save_t)
        ob = _ia(obs if obs is not None else [])
        n_obs = len(obs) if obs is not None and len(obs) else n
        n_save = int(np.asarray(save_t).size)
        out = np.zeros((max(n_save, 1), n_obs))
        o = TranOptsC(t0, t1, reltol, _dp(at), _dp(em), h0, hmin, hmax, max_newton, max_order, int(use_pcnr), newton_tol,
                      int(np.asarray(breaks).size), _dp(br), n_save, _dp(sv), len(obs) if obs is not None else 0, _ip(ob), int(newton_mode), int(step_rule))
        st = TranStatsC()
        trace = np.zeros(max(trace_cap, 1))
        ntr = C.c_int32()
        lib().port_tran(self.p, _dp(u), C.byref(o), _dp(out), C.byref(st), _dp(trace), int(trace_cap), C.byref(ntr))
        stats = {f: getattr(st, f) for f, _ in TranStatsC._fields_}
        return out[:n_save], u, stats, trace[:ntr.value]
