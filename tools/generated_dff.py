"""The DFF benchmark with the generated Verilog-A level-1 MOSFET (va_mos1l) in place of the hand-written sp_mos1 device:
same waveform?  how fast?   python tools/generated_dff.py [n_instances]   (needs a GPU)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cadnip_jl_amd import api, benchmarks as bm
from cadnip_jl_amd.structure import expand_breakpoints


def run(generated, B, fused):
    circ = bm.dff_circuit(generated=generated)
    nv = 32
    pts = list(api.ProductSweep(vdd=np.linspace(4.5, 5.5, nv), temp=np.full(max(1, B // nv), 27.0)))
    sim = api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), pts)
    st = sim.st
    sim.analyze()
    ts = np.linspace(5e-9, 7e-7, 140)
    out = None
    for rep in range(2):
        u, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=bool(fused))
        assert np.all(conv)
        sim.h.set_spec(mode="tran")
        t0 = time.time()
        out, per, stats = sim.h.tran_run(0.0, 7e-7, st.state_abstol(vntol=1e-6, iabstol=1e-9, chgtol=1e-6), 1e-4,
                                         breaks=expand_breakpoints(st.breakpoints, bm.DFF_TSPAN), save_t=ts, obs=[st.index_of("Q")], fused=fused)
        dt = time.time() - t0
    sim.close()
    print("%-12s fused=%d  n=%3d nnz=%4d  %7.2f M Newton iterations/s (%d iterations, %.1f ms, failed %d)" % (
        "generated" if generated else "hand-written", fused, st.n, st.nnz, stats["newton_iters"] / dt / 1e6, stats["newton_iters"], dt * 1e3, stats["n_failed"]))
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    a = run(False, B, 1)
    b = run(True, B, 1)
    ts = np.linspace(5e-9, 7e-7, 140)
    d = np.abs(a - b)[:, :, 0]
    # after the first clock edge (before it Q shows the latch's non-unique DC state) and outside the D / CLKN race of
    # the stimulus
    calm = (ts > 6e-8) & ((ts < 4.0e-7) | (ts > 6.2e-7))
    print("max |Q_generated - Q_hand-written| over %d corners: %.3e V on 60-400 and 620-700 ns (%.3e V anywhere); "
          "median %.3e V" % (a.shape[0], np.max(d[:, calm]), np.max(d), np.median(d)))
    worst = np.argsort(d.max(axis=0))[::-1][:6]
    print("largest differences at t =", ", ".join("%.0f ns (%.2e V, %d corners > 10 mV)" % (ts[k] * 1e9, d[:, k].max(), int((d[:, k] > 1e-2).sum())) for k in sorted(worst)))
    run(True, min(B, 1024), 0)


if __name__ == "__main__":
    main()
