"""Which fused kernel for which batch size: the DFF corner sweep at B = 1 ... 1024 with one wave per instance (k_fused2) and with a team of
four waves per instance (k_fteam), Newton mode 1.   python tools/team_scan.py [B ...]   (needs a GPU)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadnip_jl_amd import api, benchmarks as bm
from cadnip_jl_amd.structure import expand_breakpoints


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1, 32, 128, 256, 320, 384, 512, 768, 1024]
    print("%6s %14s %14s %14s" % ("B", "1 wave: ms", "2 waves: ms", "4 waves: ms"))
    for B in sizes:
        nv = min(32, B)
        pts = list(api.ProductSweep(vdd=np.linspace(4.5, 5.5, nv) if nv > 1 else np.array([5.0]), temp=np.linspace(-40, 125, max(1, B // nv)) if B // nv > 1 else np.array([27.0])))
        row = []
        for team in (0, 2, 4):
            os.environ["CADNIP_F2_TEAM"] = str(team)
            sim = api.BatchSimulator(api.MNACircuit(bm.dff_circuit(), {"vdd": 5.0}), pts)
            st = sim.st
            sim.analyze()
            best = 1e9
            for rep in range(3):
                sim.dc(abstol=1e-9, mode="tranop", fused=True)
                sim.h.set_spec(mode="tran")
                t0 = time.time()
                out, per, stats = sim.h.tran_run(0.0, 7e-7, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4, breaks=expand_breakpoints(st.breakpoints, bm.DFF_TSPAN),
                                                 save_t=np.array([7e-7]), obs=[st.index_of("Q")], fused=1, newton_mode=1)
                best = min(best, time.time() - t0)
                assert stats["n_failed"] == 0
            sim.close()
            row.append(best * 1e3)
        print("%6d %14.2f %14.2f %14.2f" % (len(pts), row[0], row[1], row[2]))


if __name__ == "__main__":
    main()
