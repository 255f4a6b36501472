"""Cycle timeline of one wave of k_fused2 over a DFF corner sweep (diagnostic build: csrc/build.sh --trace).

usage: python tools/trace_fused2.py [n_instances [newton_mode [team_wave]]]     (needs a GPU; loads libcadnip_hip_trace.so)
team_wave: trace wave `team_wave` of the first team of the team kernel (fused_team_kernel.hpp; batches of at most one instance per CU, or
CADNIP_F2_TEAM=4) instead of wave 0 of k_fused2.
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
    team = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    nv = min(32, B)
    mc = api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0})
    pts = list(api.ProductSweep(vdd=np.linspace(4.5, 5.5, nv) if nv > 1 else np.array([5.0]), temp=np.linspace(-40, 125, max(1, B // nv)) if B // nv > 1 else np.array([27.0])))
    sim = api.BatchSimulator(mc, pts)
    st = sim.st
    sim.analyze()
    sim.dc(abstol=1e-9, mode="tranop")
    lib = hip.load_library()
    s = (C.c_ulonglong * 64)(); c = (C.c_ulonglong * 64)()
    sim.h.set_spec(mode="tran")
    atol = st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    trace = (lambda reset: lib.cadnip_debug_trace_team(s, c, reset, team)) if team >= 0 else (lambda reset: lib.cadnip_debug_trace(s, c, reset))
    trace(1)
    t0 = time.time()
    out, per, stats = sim.h.tran_run(0.0, bm.DFF_TSPAN[1], atol, 1e-4, breaks=breaks, save_t=np.array([7e-7]), obs=[st.index_of("Q")], fused=2, newton_mode=mode)
    wall = time.time() - t0
    trace(0)
    rounds = c[17]
    print("instances %d  newton_mode %d  wall %.3f s  newton %d  rounds traced (%s) %d" % (len(pts), mode, wall, stats["newton_iters"], "team wave %d" % team if team >= 0 else "wave 0", rounds))
    tot = sum(s[i] for i in ORDER)
    print("%-28s %10s %8s %7s" % ("phase", "cyc/round", "calls/r", "share"))
    for i in ORDER:
        if c[i]:
            print("%-28s %10.0f %8.2f %6.1f%%" % (NAMES[i], s[i] / rounds, c[i] / rounds, 100.0 * s[i] / tot))
    print("%-28s %10.0f" % ("total", tot / rounds))


if __name__ == "__main__":
    main()
