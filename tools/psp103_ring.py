"""BASELINE.json config 5 timed on the GPU: the 9-stage PSP103 ring oscillator of benchmarks/vacask/ring (371 unknowns, 18
generated PSP103 devices evaluated with in-kernel Dual<12> arithmetic), CedarTranOp start, tspan (0, 1 us), dtmax = 50 ps
(benchmarks/vacask/ring/cedarsim/runme.jl:47-69) -- as one transient and as a batch of supply corners (Vdd of every instance
set through the V-source block's packed parameter, so no model source is needed where this runs).

    python tools/psp103_ring.py [--tspan 1e-6] [--batch 1,64,512] [--reltol 1e-3]

Reference figures for the same deck (doc/ring_oscillator_investigation.md:296-313): Cadnip 333.5 s, 242 293 Newton iterations,
1 376 us per iteration; VACASK (compiled C / OSDI) 2.28 s, 81 949 iterations, 27.8 us per iteration.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cadnip_jl_amd import api, structure as S                     # noqa: E402
from cadnip_jl_amd.structure import expand_breakpoints            # noqa: E402


def run(B, tspan, reltol, vntol=1e-6, newton_mode=1, vdd_lo=1.1, vdd_hi=1.3):
    st, x = S.load_structure(os.path.join(ROOT, "tests", "golden", "psp103_ring.npz"))
    packed = [np.repeat(x["packed%d" % i], B, axis=0) for i in range(int(x["n_packed"][0]))]
    vb = next(i for i, b in enumerate(st.blocks) if b.type == "V")
    vdd = np.linspace(vdd_lo, vdd_hi, B) if B > 1 else np.array([1.2])
    packed[vb][:, 0, 0] = vdd                                       # dc value of vdd (the only V source)
    sim = api.BatchSimulator.from_packed(st, packed, api.MNASpec(mode="tran", temp=27.0), vscale=1.2)
    sim.analyze()
    t0 = time.time()
    u, conv, dcs = sim.dc(abstol=1e-9, mode="tranop")
    t_dc = time.time() - t0
    assert conv.all(), "DC start failed"
    ts = np.linspace(0.0, tspan, 2001)
    sim.h.set_spec(mode="tran")
    t0 = time.time()
    out, per, stats = sim.h.tran_run(0.0, tspan, st.state_abstol(vntol=vntol, iabstol=1e-3 * vntol, chgtol=vntol), reltol,
                                     breaks=expand_breakpoints(st.breakpoints, (0.0, tspan)), save_t=ts, obs=[st.index_of("1")], hmax=50e-12, fused=0, newton_mode=newton_mode)
    wall = time.time() - t0
    sim.close()
    v = out[:, :, 0]
    late = ts > 0.4 * tspan
    per_ns = []
    for i in range(B):
        w = v[i, late]
        up = np.flatnonzero((w[:-1] < 0.5 * vdd[i]) & (w[1:] >= 0.5 * vdd[i]))
        per_ns.append(np.mean(np.diff(ts[late][up])) * 1e9 if len(up) > 2 else float("nan"))
    it = stats["newton_iters"]
    print("B = %4d  tspan %.0f ns  reltol %g vntol %g newton_mode %d: DC %.2f s (%d iterations), transient %.2f s, %d Newton iterations (%d accepted / %d rejected steps, "
          "%d failed instances) -> %.1f us per Newton round of the batch, %.2f us per instance-iteration; period %.3f ns at %.2f V%s" % (
              B, tspan * 1e9, reltol, vntol, newton_mode, t_dc, dcs["newton_iters"], wall, it, stats["steps_accepted"], stats["steps_rejected"], stats["n_failed"],
              wall / max(stats["launches"], 1) * 1e6, wall / max(it, 1) * 1e6, per_ns[0], vdd[0],
              (", %.3f ns at %.2f V" % (per_ns[-1], vdd[-1])) if B > 1 else ""), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tspan", type=float, default=1e-6)
    ap.add_argument("--batch", default="1,64,512")
    ap.add_argument("--reltol", type=float, default=1e-3)
    ap.add_argument("--vntol", type=float, default=1e-6, help="absolute tolerance of voltages and (scaled) charges; currents 1e-3 of it. "
                    "The reference's run uses abstol = 1e-4, reltol = 1e-2 (runme.jl:66)")
    ap.add_argument("--newton-mode", type=int, default=1, help="1 = IDA's convergence test (per-op path: a refactorisation every round), 0 = update norm < 1e-3")
    a = ap.parse_args()
    for B in [int(b) for b in a.batch.split(",")]:
        run(B, a.tspan, a.reltol, a.vntol, a.newton_mode)


if __name__ == "__main__":
    main()
