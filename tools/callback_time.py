"""Cost of ONE Newton iteration through the drop-in callback ABI at B = 1 (what a Julia host that keeps the Newton loop to itself pays per iteration):
the five entry points one after the other, cadnip_newton_step (the same kernels, one call, HIP graph), cadnip_newton_step_fused (one kernel), each
through the Python wrapper and -- to separate the wrapper from the library -- as raw ctypes calls on preallocated buffers, beside an entry
point that does nothing on the GPU (cadnip_lu_stats).   python tools/callback_time.py   (needs a GPU)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadnip_jl_amd import api, benchmarks as bm, hip


def main():
    sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), [{"vdd": 5.0, "temp": 27.0}])
    st, h = sim.st, sim.h
    sim.analyze()
    u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
    h.set_spec(mode="tran")
    rng = np.random.default_rng(0)
    du, gam, rhs = rng.random((1, st.n)), np.array([1e9]), rng.random((1, st.n))
    N = 300

    def timeit(fn):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(N):
                fn()
            best = min(best, (time.perf_counter() - t0) / N)
        return best * 1e6

    def five():
        h.rebuild(u0, 1e-8); h.residual(du, u0); h.jacobian(gam, readback=False); h.factor(); h.solve(rhs)
    print("five entry points (wrapper)                 %7.1f us" % timeit(five))
    print("cadnip_newton_step, refresh (wrapper)       %7.1f us" % timeit(lambda: h.newton_step(u0, du, gam, 1e-8)))
    print("cadnip_newton_step, kept factors (wrapper)  %7.1f us" % timeit(lambda: h.newton_step(u0, du, None, None, refresh=False)))
    print("cadnip_newton_step_fused, refresh (wrapper) %7.1f us" % timeit(lambda: h.newton_step(u0, du, gam, 1e-8, fused=True)))
    print("cadnip_newton_step_fused, kept (wrapper)    %7.1f us" % timeit(lambda: h.newton_step(u0, du, None, None, refresh=False, fused=True)))
    lib = h.lib
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    uu, dd, gg, tt = np.ascontiguousarray(u0), np.ascontiguousarray(du), np.ascontiguousarray(gam), np.array([1e-8])
    delta, nrm = np.empty_like(uu), np.empty(1)
    args = (h.h, dp(uu), dp(dd), dp(gg), dp(tt), C.c_int32(1), dp(delta), dp(nrm), None)
    print("cadnip_newton_step, refresh (raw ctypes)    %7.1f us" % timeit(lambda: lib.cadnip_newton_step(*args)))
    print("cadnip_newton_step_fused, refresh (raw)     %7.1f us" % timeit(lambda: lib.cadnip_newton_step_fused(*args)))
    args0 = (h.h, dp(uu), dp(dd), None, None, C.c_int32(0), dp(delta), dp(nrm), None)
    print("cadnip_newton_step_fused, kept (raw)        %7.1f us" % timeit(lambda: lib.cadnip_newton_step_fused(*args0)))
    v = [C.c_int32() for _ in range(5)]
    refs = [C.byref(x) for x in v]
    print("an entry point without GPU work (raw)       %7.1f us" % timeit(lambda: lib.cadnip_lu_stats(h.h, *refs)))
    sim.close()


if __name__ == "__main__":
    main()
