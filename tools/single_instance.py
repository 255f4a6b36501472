"""Latency of ONE transient on the GPU (the 10x target read for a single instance): the DFF benchmark with a
batch of 1, fused kernel and per-op kernels.   python tools/single_instance.py [newton_mode [fused_only]]   (needs a GPU;
CADNIP_F2_TEAM=0 | 2 | 4 selects the waves per instance of the fused kernel)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadnip_jl_amd import api, benchmarks as bm
from cadnip_jl_amd.structure import expand_breakpoints


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # Newton mode (1 = the benchmark's)
    for B in (1, 8):
        for fused in ((1,) if len(sys.argv) > 2 else (1, 0)):
            sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), [{"vdd": 5.0, "temp": 27.0}] * B)
            st = sim.st
            sim.analyze()
            for rep in range(2):
                u, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=bool(fused))
                sim.h.set_spec(mode="tran")
                t0 = time.time()
                out, per, stats = sim.h.tran_run(0.0, 7e-7, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4,
                                                 breaks=expand_breakpoints(st.breakpoints, bm.DFF_TSPAN), save_t=np.array([7e-7]), obs=[st.index_of("Q")], fused=fused, newton_mode=mode)
                dt = time.time() - t0
            sim.close()
            it = stats["newton_iters"] // B
            print("B = %d  %-7s  %6.1f ms per transient, %d Newton iterations each -> %.1f us per Newton iteration (wall)" % (
                B, "fused" if fused else "per-op", dt * 1e3, it, dt / it * 1e6))


if __name__ == "__main__":
    main()
