"""Cycle timeline of one wave of k_fused2 over a DFF corner sweep (diagnostic build: csrc/build.sh --trace).

usage: python tools/trace_fused2.py [n_instances [newton_mode]]     (needs a GPU; loads libcadnip_hip_trace.so)
"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cadnip_jl_amd as cj
from cadnip_jl_amd import api, benchmarks as bm, hip
from cadnip_jl_amd.structure import expand_breakpoints

hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), "libcadnip_hip_trace.so")
NAMES = {17: "round top", 0: "load u/beta + zero", 8: "block0 (mos1)", 9: "block1", 10: "block2", 11: "block3", 12: "block4", 13: "pinned R/C + source blocks", 1: "stamp sync",
         2: "r += J*u", 3: "passes after the core", 4: "passes before the core", 5: "dense core solve", 6: "backward", 7: "delta write", 16: "update/controller",
         20: " mos1: loads", 21: " mos1: limiting", 22: " mos1: g_lim stamps", 23: " mos1: junction currents", 24: " mos1: drain current",
         25: " mos1: depletion charges", 26: " mos1: current stamps", 27: " mos1: charge stamps",
         30: " update: norms", 31: " update: accept (outputs, history)", 32: " update: next step (predictor)"}
ORDER = [17, 0, 13, 20, 21, 22, 23, 24, 25, 26, 27, 8, 9, 10, 11, 12, 1, 2, 4, 5, 3, 30, 31, 32, 16]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nv = 32
    mc = api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0})
    pts = list(api.ProductSweep(vdd=np.linspace(4.5, 5.5, nv), temp=np.linspace(-40, 125, max(1, B // nv))))
    sim = api.BatchSimulator(mc, pts)
    st = sim.st
    sim.analyze()
    sim.dc(abstol=1e-9, mode="tranop")
    lib = hip.load_library()
    s = (C.c_ulonglong * 64)(); c = (C.c_ulonglong * 64)()
    sim.h.set_spec(mode="tran")
    atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    lib.cadnip_debug_trace(s, c, 1)
    t0 = time.time()
    out, per, stats = sim.h.tran_run(0.0, bm.DFF_TSPAN[1], atol, 1e-4, breaks=breaks, save_t=np.array([7e-7]), obs=[st.index_of("Q")], fused=2, newton_mode=mode)
    wall = time.time() - t0
    lib.cadnip_debug_trace(s, c, 0)
    rounds = c[17]
    print("instances %d  newton_mode %d  wall %.3f s  newton %d  rounds traced (wave 0) %d" % (len(pts), mode, wall, stats["newton_iters"], rounds))
    tot = sum(s[i] for i in ORDER)
    print("%-28s %10s %8s %7s" % ("phase", "cyc/round", "calls/r", "share"))
    for i in ORDER:
        if c[i]:
            print("%-28s %10.0f %8.2f %6.1f%%" % (NAMES[i], s[i] / rounds, c[i] / rounds, 100.0 * s[i] / tot))
    print("%-28s %10.0f" % ("total", tot / rounds))


if __name__ == "__main__":
    main()
