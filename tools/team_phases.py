"""Where one Newton iteration of the team kernel (csrc/fused_team_kernel.hpp) spends its time: the STEP-mode kernel repeats the iteration at
the DFF's operating point inside ONE launch (cadnip_debug_step_time), with phases left out one at a time; differences are the phases.
(The update / step controller is not part of a STEP iteration: its cost is what a transient's round takes beyond the sum below.)

    python tools/team_phases.py     (needs a GPU)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadnip_jl_amd import api, benchmarks as bm


def main():
    sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), [{"vdd": 5.0, "temp": 27.0}])
    st, h = sim.st, sim.h
    sim.analyze()
    u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
    h.set_spec(mode="tran")
    rng = np.random.default_rng(0)
    du = (rng.random((1, st.n)) - 0.5) * 1e3
    # put the state on the device through one ordinary call, leaving factors for the kept rounds
    h.newton_step(u0, du, np.array([1e9]), 1e-8, refresh=True, fused=True)
    ms = C.c_double()
    reps = 2000

    def t(refresh, skip):
        best = 1e9
        for _ in range(3):
            rc = h.lib.cadnip_debug_step_time(h.h, C.c_int32(refresh), C.c_int32(reps), C.c_int32(skip), C.byref(ms))
            assert rc == 0, rc
            best = min(best, ms.value)
        return best * 1e3 / reps        # us per iteration

    for refresh, name in ((0, "kept factors"), (1, "refactorisation")):
        full = t(refresh, 0)
        print("%s: %.2f us per iteration (stamping, sums, linear solve; no update)" % (name, full))
        for skip, what in ((1, "stamping"), (2, "adding the waves' sums"), (4, "linear-solve steps"), (8, "dense core"), (15, "everything (loop, barriers, residual norm, output)")):
            v = t(refresh, skip)
            print("    without %-52s %6.2f us   -> %5.2f us" % (what, v, full - v if skip != 15 else v))
    sim.close()


if __name__ == "__main__":
    main()
