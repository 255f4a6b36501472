"""Throughput of the fused kernel's full device-set variant on circuits that are not lean (diodes, generated Verilog-A
modules, unpaired sp_mos1): python tools/variant_bench.py [n_instances]   (needs a GPU; CADNIP_F2_WPB caps waves / workgroup)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cadnip_jl_amd as cj
from cadnip_jl_amd import api
from tests import circuits as tc
from tests.test_gpu_tran_parity import _meyer_inverter, _nonlinear_tran


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cases = {"va_zoo": (tc.va_zoo, (0.0, 1e-6)), "va_limited": (tc.va_limited, (0.0, 2e-6)), "meyer_inverter_rd": (_meyer_inverter, (0.0, 2e-8)),
             "nonlinear (D, DCAP, SMOS)": (_nonlinear_tran, (0.0, 2e-6))}
    for name, (mk, tspan) in cases.items():
        sim = api.BatchSimulator(api.MNACircuit(mk(), {}), [{}] * B)
        st = sim.st
        sim.analyze()
        for rep in range(2):
            u, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
            t0 = time.time()
            out, per, stats = sim.tran(tspan, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4, np.array([tspan[1]]), obs=[0], fused=1)
            dt = time.time() - t0
        sim.close()
        print("%-28s n=%3d  %8.2f M Newton iterations/s  (%d iterations, %.1f ms, failed %d)" % (name, st.n, stats["newton_iters"] / dt / 1e6, stats["newton_iters"], dt * 1e3, stats["n_failed"]))


if __name__ == "__main__":
    main()
