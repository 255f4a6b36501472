"""Full Newton (mode 0) against IDA-style Jacobian reuse (mode 1, csrc/tran_ctrl.hpp) on the benchmark sweep: Newton iterations, steps,
wall time and iterations / s of the fused kernel, one instance and the 4096-corner batch.   python tools/newton_mode.py [B ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cadnip_jl_amd import api, benchmarks as bm, sweep_shard                 # noqa: E402
from cadnip_jl_amd.structure import expand_breakpoints                       # noqa: E402


def run(B):
    pts = sweep_shard.rank_points(B, 0, 1)[0] if B > 1 else [{"vdd": 5.0, "temp": 27.0}]
    sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), pts)
    st = sim.st
    sim.analyze()
    atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    for mode in (0, 1, 0, 1):
        u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
        sim.h.set_spec(mode="tran")
        t0 = time.perf_counter()
        out, per, stats = sim.h.tran_run(0.0, 7e-7, atol, 1e-4, breaks=breaks, save_t=[7e-7], obs=[st.index_of("Q")], fused=2, newton_mode=mode)
        wall = time.perf_counter() - t0
        print("B = %5d  newton_mode %d: %9d Newton iterations, %8d accepted / %7d rejected steps, %d failed; %.1f ms, %.2f M iterations/s, %.2f us per "
              "iteration of one instance; Q(700 ns) in [%.4f, %.4f]" % (B, mode, stats["newton_iters"], stats["steps_accepted"], stats["steps_rejected"], stats["n_failed"],
                                                                          wall * 1e3, stats["newton_iters"] / wall / 1e6, wall / (stats["newton_iters"] / B) * 1e6,
                                                                          out[:, 0, 0].min(), out[:, 0, 0].max()), flush=True)
    sim.close()


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:]] or [1, 1024, 4096]:
        run(B)
