#!/usr/bin/env python3
"""bench.py -- Newton iterations/sec on the synthetic gf180 DFF transient (BASELINE.json metric).

One "step" = one complete tran! of the per-GPU sweep batch: CedarTranOp-style DC initialisation (:tranop, PCNR Newton) +
the 0-700 ns transient of every resident sweep instance, results gathered (RCCL all_gather, inside the timed region --
the only collective of the path).

Two workloads are timed in every run and printed in the same JSON line (rank 0):
  value        : WEAK scaling -- every rank integrates ``--instances`` corner points (default 4096 = BASELINE.json config 4's
                 32-value Vdd axis x 128 temperatures); with N ranks the temperature axis is refined N-fold over the same
                 range, so N GPUs process N x 4096 independent transients.
  strong_1024  : STRONG scaling, BASELINE.json config 4 as worded -- the 32 x 32 = 1024-point Vdd x temp sweep split into
                 contiguous blocks of 1024 / N points per GPU.

Further objects of the line (rank 0; the ones marked N=1 are measured only without other ranks):
  roofline     : the dominant kernel against the roof that bounds it.  The fused Newton kernel keeps its working set in LDS
                 and registers (65 B of HBM traffic per Newton round), so its roof is vector-instruction issue:
                 bound "valu_issue", achieved = SIMD cycles/s with a vector instruction in flight, peak = SIMDs x clock,
                 frac = VALU-busy fraction, all from a LIVE `rocprofv3 --pmc` pass of this same workload (a child process;
                 N=1).  The HBM view of the same kernel (counter bytes / launch duration / 8 TB/s) is `roofline.hbm`.  When
                 rocprofv3 is not available the numbers of the committed profile are used and say so (`source`).
  stamp_kernel : north_star's evidence line -- the per-op stamping kernel of the benchmark's device type (sp_mos1: stamp +
                 segmented reduction into CSR, csrc/stamp_csr.hip) at B = 8192, algorithmic bytes / its
                 launch duration (HIP events over back-to-back launches) against the 8 TB/s HBM peak (N=1).
  single_instance_us_per_iter, callback_us_per_iter : one DFF transient alone on the GPU, and the host-pointer callback
                 sequence rebuild -> residual -> jacobian -> factor -> solve a Julia integrator would drive (N=1).
  cpu_baseline : the oracle's C++ port (oracle/cpu_port.cpp) on a bounded sample of the same workload on the host cores.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9          # same guide: max clock 2400 MHz
SIMDS = 256 * 4           # 256 CUs x 4 SIMDs
ABSTOL = dict(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
RELTOL = 1e-4
COMMITTED_PMC = "r02h_fused_B4096_pmc_summary.json"   # fallback when no live counter pass is possible (profiles/)
SQ_COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
               "SQ_INSTS_SALU", "SQ_INSTS_LDS"]
CALIB = "k_calib_copy_f64"
# the reference's own per-Newton-iteration costs (it cannot be run here: Julia is absent), doc/ring_oscillator_investigation.md:228-234
REFERENCE_PUBLISHED = {"source": "doc/ring_oscillator_investigation.md:228-234 (9-stage ring, reference CPU path, per Newton iteration)",
                       "mos1_n11_us": 49.5, "bsim4_n191_us": 793.0, "psp103_n371_us": [1376.0, 2074.0]}


def algorithmic_bytes(st, B, nnz_lu):
    """Per-launch algorithmic bytes of each per-op kernel for a batch of B instances (SURVEY.md section 8d / DESIGN.md
    section 5: every array counted once per required read or write; the per-slot contributions count although the
    stamping kernel keeps them in LDS -- SURVEY: "may stay in LDS; still counted")."""
    out = {}
    names = {"R": "resistor", "C": "capacitor", "L": "inductor", "V": "vsource", "I": "isource", "E": "vcvs", "G": "vccs", "H": "ccvs",
             "F": "cccs", "D": "diode", "DCAP": "diodecap", "SMOS": "simplemos", "MOS1": "mos1"}
    touched = stamp_targets(st)
    for k, blk in enumerate(st.blocks):
        if blk.type.startswith("VA:"):
            # a generated Verilog-A model: parameter rows, node indices, staged contributions + their maps, the unknowns, the reduced words
            n_slots = blk.n_g + blk.n_c + blk.n_b
            per_dev = 8 * blk.n_par + 4 * blk.nodes.shape[0] + (8 + 4) * n_slots
            key = "stamp_va_" + blk.type[3:]
            out[key] = out.get(key, 0) + B * (blk.count * per_dev + 8 * st.n + 8 * touched[k])
            continue
        if blk.type not in names:
            continue
        n_slots = blk.n_g + blk.n_c + blk.n_b
        per_dev = 8 * blk.n_par + 4 * blk.nodes.shape[0] + (8 + 4) * n_slots        # params, node indices, contributions + their maps
        if blk.type in ("D", "MOS1"):
            per_dev += 8 * (1 if blk.type == "D" else 4)                               # limit_w
        out["stamp_" + names[blk.type]] = B * (blk.count * per_dev + 8 * st.n + 8 * touched[k])   # + u read, reduced G / C / b written
    out["residual"] = B * (2 * 8 * st.nnz + 8 * 4 * st.n) + 4 * (st.nnz + st.n + 1)
    out["lu_factor_solve"] = B * (8 * 2 * st.nnz + 8 * nnz_lu + 8 * 2 * st.n)
    out["tran_update"] = B * 8 * 8 * st.n
    out["B_iter_per_instance"] = sum(v for kk, v in out.items()) // B
    return out


def stamp_targets(st):
    """Per device block: the number of words of G, C and b its stamping kernel writes (targets of its reduction)."""
    res = []
    for blk in st.blocks:
        n = 0
        for ptr, slots, base, nk in ((st.g_ptr, st.g_slots, blk.g_base, blk.n_g), (st.c_ptr, st.c_slots, blk.c_base, blk.n_c),
                                     (st.b_ptr, st.b_slots, blk.b_base, blk.n_b)):
            sl = np.asarray(slots)
            own = (sl >= base) & (sl < base + nk * blk.count)
            tgt = np.repeat(np.arange(len(ptr) - 1), np.diff(np.asarray(ptr)))
            n += len(np.unique(tgt[own]))
        res.append(n)
    return res


# ---- rocprofv3 child passes ---------------------------------------------------------------------------------------------------
def _rocprof(counters, probe_args, timeout=240):
    """One `rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --probe ...` pass; returns (probe JSON, per-kernel
    {counter: total, "calls": n, "ns": summed kernel durations}) or None.  Counter passes never carry other trace domains."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="cadnip_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "-d", tmp, "-o", "run", "--output-format", "csv", "--",
                                      sys.executable, os.path.join(ROOT, "bench.py"), "--probe"] + probe_args
    try:
        p = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        probe = None
        for line in p.stdout.splitlines():
            if line.startswith("{") and '"probe"' in line:
                probe = json.loads(line)
        if p.returncode != 0 or probe is None:
            return None
        acc = {}
        for fn in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
            seen = set()
            for row in csv.DictReader(open(fn)):
                k = acc.setdefault(row["Kernel_Name"], {"calls": 0, "ns": 0.0})
                k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    k["calls"] += 1
                    k["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        return probe, acc
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _find(acc, *needles):
    for k, v in acc.items():
        if all(nd in k for nd in needles):
            return v
    return None


def live_counters(B, fused, newton_mode):
    """SQ pass + the two HBM passes (FETCH_SIZE, WRITE_SIZE in separate runs, calibrated with the 1 GiB copy kernel as the
    microarchitecture guide prescribes) of the benchmark workload and of the per-op stamping kernel at B = 8192."""
    args = ["--instances", str(B), "--fused", str(fused), "--newton-mode", str(newton_mode)]
    out = {}
    sq = _rocprof(SQ_COUNTERS, args)
    if sq is None:
        return None
    probe, acc = sq
    f = _find(acc, "k_fused2<", ", false, ") or _find(acc, "k_fteam<")          # (a batch of at most one instance per CU runs in the team kernel)
    if f is not None and probe.get("newton_iters"):
        it = probe["newton_iters"]
        wc = f["SQ_WAVE_CYCLES"]
        out["fused"] = {"kernel_ns": f["ns"], "calls": f["calls"], "newton_iters": it,
                        "valu_busy_simd_cycles": 4.0 * f["SQ_ACTIVE_INST_VALU"],
                        "per_newton_round": {"valu_insts": f["SQ_INSTS_VALU"] / it, "salu_insts": f["SQ_INSTS_SALU"] / it,
                                             "lds_insts": f["SQ_INSTS_LDS"] / it, "wave_cycles": 4.0 * wc / it},
                        "wave_time_shares": {"waiting_on_waitcnt": f["SQ_WAIT_ANY"] / wc, "issue_stalled": f["SQ_WAIT_INST_ANY"] / wc,
                                             "issuing": f["SQ_ACTIVE_INST_ANY"] / wc, "issuing_valu": f["SQ_ACTIVE_INST_VALU"] / wc}}
    hbm = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        r = _rocprof([ctr], args + ["--calib-copy", "1024"])
        if r is None:
            return out or None
        hbm[ctr] = r[1]
    cal_f, cal_w = _find(hbm["FETCH_SIZE"], CALIB), _find(hbm["WRITE_SIZE"], CALIB)
    if cal_f and cal_w:
        # the copy streams 1 GiB in and 1 GiB out per call; counters are in KiB.  On gfx950 FETCH_SIZE reads half the true
        # bytes (correction ~2, MI355X_MICROARCH.md HBM section); both corrections are measured, not assumed.
        fc = 1024.0 * 1024.0 / (cal_f["FETCH_SIZE"] / cal_f["calls"])
        wc_ = 1024.0 * 1024.0 / (cal_w["WRITE_SIZE"] / cal_w["calls"])
        out["calibration"] = {"fetch_correction": round(fc, 4), "write_correction": round(wc_, 4)}
        for key, needles in (("fused", ("k_fused2<", ", false, ")), ("fused", ("k_fteam<",)), ("stamp", ("k_stamp_csr<12,",))):
            a, b = _find(hbm["FETCH_SIZE"], *needles), _find(hbm["WRITE_SIZE"], *needles)
            if a and b and a["calls"] and b["calls"]:
                byt = (a["FETCH_SIZE"] / a["calls"] * fc + b["WRITE_SIZE"] / b["calls"] * wc_) * 1024.0
                out.setdefault(key, {})["hbm_bytes_per_launch"] = byt
                out[key]["hbm_pass_avg_ns"] = a["ns"] / a["calls"]
    return out


def probe_main(args):
    """Child of a rocprofv3 pass: one step of the benchmark workload, then the per-op stamping kernel at B = 8192."""
    import cadnip_jl_amd  # noqa: F401
    from cadnip_jl_amd import api, benchmarks as bm, sweep_shard
    from cadnip_jl_amd.structure import expand_breakpoints
    circ = bm.dff_circuit()
    pts, _ = sweep_shard.rank_points(args.instances, 0, 1)
    sim = api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), pts)
    st = sim.st
    sim.analyze()
    if args.calib_copy:
        sim.h.debug_copy(args.calib_copy * (1 << 20) // 8, 4)
    atol = st.state_abstol(**ABSTOL)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    u0, conv, dcs = sim.dc(abstol=1e-9, mode="tranop", fused=bool(args.fused))
    if not np.all(conv):                       # counters of a workload that did not run as benchmarked are worth nothing
        raise RuntimeError("probe: DC initialisation failed for %d of %d instances (%s)" % (int((~np.asarray(conv, dtype=bool)).sum()), len(conv), dcs))
    sim.h.set_spec(mode="tran")
    _, _, stats = sim.h.tran_run(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, RELTOL, breaks=breaks, save_t=np.linspace(*bm.DFF_TSPAN, 71),
                                 obs=[st.index_of("Q")], fused=int(args.fused), newton_mode=args.newton_mode)
    sim.close()
    stamp = stamp_kernel_leg(circ, reps=20)
    print(json.dumps({"probe": 1, "newton_iters": int(stats["newton_iters"]), "launches": int(stats["launches"]), "stamp": stamp}))


def stamp_kernel_leg(circ, B=8192, reps=50):
    """The per-op stamping kernel of the benchmark's device type at B instances, at the DC operating points of the sweep:
    average duration of back-to-back launches (HIP events on the launching stream)."""
    from cadnip_jl_amd import api, sweep_shard
    pts, _ = sweep_shard.rank_points(B, 0, 1)
    sim = api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), pts)
    try:
        st = sim.st
        sim.analyze()
        u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
        sim.h.set_spec(mode="tran")
        sim.h.rebuild(u0, 5.05e-8)
        blk = next(k for k, b in enumerate(st.blocks) if b.type == "MOS1")
        sim.h.stamp_time(blk, 5)
        ms = sim.h.stamp_time(blk, reps)
        ms_all = sim.h.stamp_time(-1, reps)
        ab = algorithmic_bytes(st, B, sim.h.lu_stats()["nnz_lu"])
        us = 1e3 * ms / reps
        return {"name": "k_stamp_csr<12> (sp_mos1: stamp + segmented reduction into CSR)", "B": B, "avg_us": round(us, 3),
                "algorithmic_bytes": int(ab["stamp_mos1"]), "achieved_GBps": round(ab["stamp_mos1"] / us / 1e3, 1),
                "frac": round(ab["stamp_mos1"] / us / 1e3 / HBM_PEAK_GBS, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "bound": "hbm",
                "restamp_all_blocks_avg_us": round(1e3 * ms_all / reps, 3), "traffic": None}
    finally:
        sim.close()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--instances", type=int, default=4096, help="weak-scaling sweep instances per GPU (32 Vdd x instances/32 temps)")
    ap.add_argument("--total-instances", type=int, default=1024, help="strong-scaling sweep: this many points split over the GPUs (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=1536, help="corner points timed on the host for cpu_baseline")
    ap.add_argument("--fused", type=int, default=int(os.environ.get("CADNIP_FUSED", "2")),
                    help="0 = one kernel per op (the kernels behind the callback ABI), non-zero = fused Newton kernel (default)")
    ap.add_argument("--newton-mode", type=int, default=-1, help="1 = the nonlinear iteration as the reference's IDA runs it (Jacobian reuse, rate test: "
                    "csrc/tran_ctrl.hpp; fused kernel and CPU port), 0 = full Newton with a fixed update tolerance; default: 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the rocprofv3 child passes (roofline falls back to the committed profile)")
    ap.add_argument("--no-extras", action="store_true", help="skip stamp_kernel / single-instance / callback legs")
    ap.add_argument("--probe", action="store_true", help="internal: child of a rocprofv3 pass")
    ap.add_argument("--calib-copy", type=int, default=0, help="also run the fp64 calibration copy of this many MiB (HBM counter passes)")
    args = ap.parse_args(argv)
    if args.newton_mode < 0:
        args.newton_mode = 1          # (per-op path: the convergence test only -- it refactors every round)
    if args.probe:
        return probe_main(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    result = run(args, rank, world, local_rank, dist if world > 1 else None, dev,
                 sync=torch.cuda.synchronize, reduce_max=lambda x: _allreduce(torch, dist, dev, x, "max"),
                 reduce_sum=lambda x: _allreduce(torch, dist, dev, x, "sum"))
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def _allreduce(torch, dist, dev, x, op):
    if isinstance(x, float):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
    else:
        t = torch.tensor([x], device=dev, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item()) if isinstance(x, float) else int(t.item())


def make_simulator(circ, pts, device):
    from cadnip_jl_amd import api
    return api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), pts, device=device)


def run(args, rank, world, local_rank, dist, dev, sync, reduce_max, reduce_sum, simulator=make_simulator):
    """Everything between process-group set-up and printing; ``simulator`` is replaceable so that the control flow (every
    rank takes part in every collective, rank 0 alone runs the single-GPU extras) can be driven on CPU ranks in the tests."""
    import cadnip_jl_amd  # noqa: F401
    from cadnip_jl_amd import benchmarks as bm, sweep_shard
    from cadnip_jl_amd.structure import expand_breakpoints

    circ = bm.dff_circuit()
    save_t = np.linspace(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], 71)

    def workload(pts):
        sim = simulator(circ, pts, local_rank)
        st = sim.st
        sim.analyze()
        atol = st.state_abstol(**ABSTOL)
        breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
        obs = [st.index_of("Q")]

        def one_step(gather=True, newton_mode=None):
            u0, conv, dcs = sim.dc(abstol=1e-9, mode="tranop", fused=bool(args.fused))
            if not np.all(conv):
                raise RuntimeError("DC initialisation failed on rank %d: %d of %d instances unconverged (%s)" % (rank, int((~np.asarray(conv, dtype=bool)).sum()), len(conv), dcs))
            sim.h.set_spec(mode="tran")
            out, per, stats = sim.h.tran_run(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, RELTOL, breaks=breaks, save_t=save_t, obs=obs,
                                             fused=int(args.fused), newton_mode=args.newton_mode if newton_mode is None else newton_mode)
            if stats["n_failed"]:
                raise RuntimeError("%d transient(s) failed on rank %d" % (stats["n_failed"], rank))
            if gather and world > 1:   # final gather of the result blocks over RCCL / xGMI: every rank calls it, or none
                sweep_shard.gather_blocks(out, world, dev)
            return stats["newton_iters"] + dcs["newton_iters"], out, stats
        return sim, one_step

    def timed(one_step, steps, warmup):
        for _ in range(warmup):
            one_step()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        iters, last = 0, None
        for _ in range(steps):
            n_it, _, last = one_step()
            iters += n_it
        sync()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            elapsed = reduce_max(float(elapsed))
            iters = reduce_sum(int(iters))
        return iters, elapsed, last

    # ---- weak scaling: the headline value -------------------------------------------------------------------------------------
    pts, _ = sweep_shard.rank_points(args.instances, rank, world)
    B = len(pts)
    sim, one_step = workload(pts)
    st = sim.st
    iters, elapsed, last = timed(one_step, args.steps, args.warmup)

    # ---- strong scaling: BASELINE.json config 4 as worded (1024 points over the GPUs), every rank takes part ---------------------
    strong = None
    if args.total_instances > 0:
        spts, _ = sweep_shard.strong_points(args.total_instances, rank, world)
        ssim, s_step = workload(spts)
        s_iters, s_elapsed, _ = timed(s_step, max(1, min(args.steps, 3)), 1)
        ssim.close()
        strong = {"value": round(s_iters / s_elapsed, 1), "unit": "newton_iters/s", "ms_per_step": round(1e3 * s_elapsed / max(1, min(args.steps, 3)), 3),
                  "instances_total": args.total_instances, "instances_per_gpu": len(spts), "scaling": "strong"}

    result = None
    if rank == 0:
        # ---- per-kernel launch durations: one extra, event-timed step on THIS rank only (no collective inside) -------------------
        sim.h.profile(True)
        _, _, pstats = one_step(gather=False)
        prof = sim.h.profile_read()
        sim.h.profile(False)
        nnz_lu = sim.h.lu_stats()["nnz_lu"]
        ab = algorithmic_bytes(st, B, nnz_lu)
        dom = max(prof.items(), key=lambda kv: kv[1][0])
        kernels = {k: {"ms_total": round(v[0], 3), "calls": int(v[1]), "avg_us": round(1e3 * v[0] / max(v[1], 1), 3)} for k, v in prof.items()}
        name = dom[0]
        avg_s = dom[1][0] / max(dom[1][1], 1) * 1e-3
        value = iters / elapsed
        roof = {"kernel": name, "avg_launch_us": round(avg_s * 1e6, 3), "kernels": kernels, "B_iter_per_instance": int(ab["B_iter_per_instance"])}
        live = None
        if world == 1 and not args.no_live_pmc:
            live = live_counters(B, int(args.fused), args.newton_mode)
        if name.startswith("fused"):
            iters_per_launch = pstats["newton_iters"] / max(dom[1][1], 1)
            roof.update({"bound": "valu_issue", "unit": "G SIMD-cycles/s", "peak": round(SIMDS * CLOCK_HZ / 1e9, 1),
                         "newton_iters_per_launch": round(iters_per_launch, 1)})
            src = None
            if live and "fused" in live and "valu_busy_simd_cycles" in live["fused"]:
                f = live["fused"]
                roof["achieved"] = round(f["valu_busy_simd_cycles"] / f["kernel_ns"], 2)     # SIMD cycles with a VALU instruction in flight per ns
                roof["per_newton_round"] = {k: round(v, 1) for k, v in f["per_newton_round"].items()}
                roof["wave_time_shares"] = {k: round(v, 3) for k, v in f["wave_time_shares"].items()}
                roof["valu_insts_per_s"] = round(f["per_newton_round"]["valu_insts"] * value / 1e9, 2)   # G wave-instructions/s at the timed rate
                src = "live rocprofv3 --pmc pass of this workload (child process)"
                if "hbm_bytes_per_launch" in f:
                    roof["traffic"] = int(f["hbm_bytes_per_launch"])
                    roof["hbm"] = {"bound": "hbm", "achieved": round(f["hbm_bytes_per_launch"] / avg_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(f["hbm_bytes_per_launch"] / avg_s / 1e9 / HBM_PEAK_GBS, 6),
                                   "bytes_per_newton_round": round(f["hbm_bytes_per_launch"] / max(iters_per_launch, 1), 1),
                                   "calibration": live.get("calibration")}
            else:
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", COMMITTED_PMC)))["fused2_newton"]
                    roof["achieved"] = round(pmc["valu_busy_simd_cycles_per_ns"], 2)
                    roof["per_newton_round"] = pmc["per_newton_round"]
                    roof["wave_time_shares"] = pmc["wave_time_shares"]
                    roof["traffic"] = pmc.get("hbm_bytes_per_launch")
                    src = "committed profile profiles/%s (no live counter pass in this run)" % COMMITTED_PMC
                except (OSError, ValueError, KeyError):
                    roof["achieved"] = None
            roof["source"] = src
            roof["frac"] = None if roof.get("achieved") is None else round(roof["achieved"] / roof["peak"], 4)
            roof["algorithmic_GBps_info"] = round(ab["B_iter_per_instance"] * iters_per_launch / avg_s / 1e9, 1)
            roof["note"] = ("the fused kernel holds J, rhs, u and the history in LDS / registers: HBM is not its roof (see roofline.hbm); frac = share of "
                            "SIMD cycles with a vector instruction in flight (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel time x 2.4 GHz)); "
                            "algorithmic_GBps_info is SURVEY 8d's B_iter x iterations / time, a bytes-avoided figure, not a utilisation")
        else:
            per_launch = ab.get(name)
            roof.update({"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": None, "source": "HIP events, algorithmic bytes (SURVEY 8d)"})
            roof["achieved"] = None if per_launch is None else round(per_launch / avg_s / 1e9, 3)
            roof["frac"] = None if per_launch is None else round(per_launch / avg_s / 1e9 / HBM_PEAK_GBS, 5)

        extras = {}
        if args.newton_mode and int(args.fused):
            # the same sweep with full Newton (every round restamps and refactors, converged at update norm < 1e-3): round 1's iteration
            fn = []
            for _ in range(2):
                sync(); t0 = time.perf_counter()
                n_it, _, fst = one_step(gather=False, newton_mode=0)
                sync(); fn.append((time.perf_counter() - t0, n_it, fst))
            dt, n_it, fst = min(fn, key=lambda x: x[0])
            extras["full_newton"] = {"value": round(n_it / dt, 1), "unit": "newton_iters/s", "ms_per_step": round(1e3 * dt, 3), "newton_iters_per_step": int(n_it),
                                     "accepted": int(fst["steps_accepted"]), "rejected": int(fst["steps_rejected"]),
                                     "note": "newton_mode 0 on the same batch: what newton_mode 1 (IDA's Jacobian reuse and rate test) is compared with"}
        if world == 1 and not args.no_extras:
            # ---- north_star's stamping-kernel line, single-instance latency, callback sequence ------------------------------------
            sk = stamp_kernel_leg(circ)
            # north_star: "rocprof achieved-HBM-GB/s on the stamping kernel against the chip's peak" -- `frac` is the COUNTER figure (HBM bytes the
            # kernel really moved / its duration / 8 TB/s) whenever a counter pass ran; SURVEY 8d's algorithmic bytes (which count the per-slot
            # contributions although they never leave LDS) stay beside it as frac_algorithmic
            sk["frac_algorithmic"] = sk["frac"]
            sk["frac_source"] = "algorithmic bytes (no counter pass in this run)"
            if live and "stamp" in live and "hbm_bytes_per_launch" in live["stamp"]:
                sk["traffic"] = int(live["stamp"]["hbm_bytes_per_launch"])
                sk["traffic_GBps"] = round(live["stamp"]["hbm_bytes_per_launch"] / (sk["avg_us"] * 1e3), 1)
                sk["frac"] = round(sk["traffic_GBps"] / HBM_PEAK_GBS, 4)
                sk["frac_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (calibrated)"
            extras["stamp_kernel"] = sk
            extras.update(single_and_callback(circ, local_rank, args))
            extras["psp103_ring"] = psp103_ring_leg(local_rank)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, circ, sim, pts, save_t)
        result = {
            "metric": "newton_iters_per_sec", "value": round(value, 1), "unit": "newton_iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "gf180 DFF transient 0-700 ns (synthetic sp_mos1 level-1 card, 30 MOSFETs, n=%d, nnz=%d), "
                                   "%d Vdd x temp corner instances per GPU (state resident in HBM; one 8-instance workgroup per CU in flight, the rest "
                                   "handed out by the kernel's instance queue); step = DC init + full transient of the batch" % (st.n, st.nnz, B),
                       "instances_per_gpu": B, "instances_total": B * world, "abstol": ABSTOL, "reltol": RELTOL,
                       "fused": int(args.fused), "newton_mode": int(args.newton_mode), "newton_iters_per_step": int(iters // max(args.steps, 1)),
                       "launches_last_step": int(last["launches"]), "accepted_last_step": int(last["steps_accepted"]),
                       "rejected_last_step": int(last["steps_rejected"])},
            "transients_per_s": round(B * world * args.steps / elapsed, 1),
            "newton_iters_per_accepted_step": round(last["newton_iters"] / max(last["steps_accepted"], 1), 3),
            "rejected_step_share": round(last["steps_rejected"] / max(last["steps_accepted"] + last["steps_rejected"], 1), 4),
            "value_note": "Newton iterations of rejected steps (rejected_step_share of all steps) and of the DC initialisation are counted in `value`: it is the kernel's iteration rate; "
                          "transients_per_s is the work rate",
            "strong_1024": strong, "roofline": roof, "cpu_baseline": cpu, "reference_published": REFERENCE_PUBLISHED,
        }
        result.update(extras)
    sim.close()
    return result


def psp103_ring_leg(device):
    """BASELINE.json config 5 beside the headline number: the reference's 9-stage PSP103 ring oscillator (n = 371, generated PSP103 evaluated
    on tangent lanes, per-op path), a 20 ns slice of its 1 us run with the reference's tolerances (runme.jl:66), alone and as 256 supply
    corners.  Structure and parameters come from the committed fixture (the model source is not on the GPU box)."""
    from cadnip_jl_amd import api, structure as S
    from cadnip_jl_amd.structure import expand_breakpoints
    out = {"workload": "benchmarks/vacask/ring: 9 stages, 18 PSP103 devices, n = 371; CedarTranOp start, 20 ns slice, dtmax 50 ps, abstol 1e-4, reltol 1e-2, IDA's convergence test (newton_mode 1; the per-op path refactors every round)",
           "reference_us_per_iter": 1376.0, "vacask_us_per_iter": 27.8, "reference_source": "doc/ring_oscillator_investigation.md:299-313"}
    try:
        st, x = S.load_structure(os.path.join(ROOT, "tests", "golden", "psp103_ring.npz"))
        for B in (1, 256):
            packed = [np.repeat(x["packed%d" % i], B, axis=0) for i in range(int(x["n_packed"][0]))]
            vb = next(i for i, b in enumerate(st.blocks) if b.type == "V")
            packed[vb][:, 0, 0] = np.linspace(1.1, 1.3, B) if B > 1 else 1.2
            sim = api.BatchSimulator.from_packed(st, packed, api.MNASpec(mode="tran", temp=27.0), device=device, vscale=1.2)
            try:
                sim.analyze()
                nnz_lu = sim.h.lu_stats()["nnz_lu"]
                u, conv, _ = sim.dc(abstol=1e-9, mode="tranop")
                sim.h.set_spec(mode="tran")
                t1 = 20e-9
                _, _, stats = sim.h.tran_run(0.0, t1, st.state_abstol(vntol=1e-4, iabstol=1e-7, chgtol=1e-4), 1e-2, breaks=expand_breakpoints(st.breakpoints, (0.0, t1)),
                                             save_t=np.array([t1]), obs=[st.index_of("1")], hmax=50e-12, fused=0, newton_mode=1)
            finally:
                sim.close()
            out["B%d" % B] = {"newton_iters": int(stats["newton_iters"]), "wall_s": round(stats["wall_seconds"], 3), "failed": int(stats["n_failed"]) + int((~conv).sum()),
                              "us_per_instance_iter": round(1e6 * stats["wall_seconds"] / max(stats["newton_iters"], 1), 3),
                              "us_per_newton_round": round(1e6 * stats["wall_seconds"] / max(stats["launches"], 1), 1)}
            # SURVEY 8d's bytes for this circuit (per instance and Newton iteration: every array of the per-op kernels counted once per required
            # read or write) against the HBM peak: the per-op path's roofline line for config 5
            b_iter = algorithmic_bytes(st, 1, nnz_lu)["B_iter_per_instance"]
            gbps = b_iter * stats["newton_iters"] / max(stats["wall_seconds"], 1e-12) / 1e9
            out["B_iter_per_instance"] = int(b_iter)
            out["B%d" % B]["roofline"] = {"bound": "hbm", "achieved": round(gbps, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 5),
                                          "note": "algorithmic bytes (B_iter_per_instance x Newton iterations) / wall time of the slice; a single ring is a latency problem (one PSP103 evaluation is ~10 k dependent vector instructions), the batch is bound by that evaluation's instruction stream, not by HBM"}
    except Exception as e:       # the headline line must not depend on this leg
        out["error"] = "%s: %s" % (type(e).__name__, e)
    return out


def single_and_callback(circ, device, args):
    """One DFF transient alone (configs 2 / 3 are single transients), and the drop-in callback sequence with host pointers."""
    from cadnip_jl_amd import api, benchmarks as bm
    from cadnip_jl_amd.structure import expand_breakpoints
    out = {}
    sim = api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), [{"vdd": 5.0, "temp": 27.0}], device=device)
    try:
        st = sim.st
        sim.analyze()
        atol = st.state_abstol(**ABSTOL)
        breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
        best = None
        for _ in range(3):
            u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
            sim.h.set_spec(mode="tran")
            _, _, stats = sim.h.tran_run(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, RELTOL, breaks=breaks, save_t=np.array([bm.DFF_TSPAN[1]]),
                                         obs=[st.index_of("Q")], fused=int(args.fused) or 2, newton_mode=args.newton_mode if int(args.fused) else 1)
            us = 1e6 * stats["wall_seconds"] / max(stats["newton_iters"], 1)
            out["single_instance_ms_per_transient"] = round(1e3 * stats["wall_seconds"], 3) if best is None or us < best else out["single_instance_ms_per_transient"]
            best = us if best is None else min(best, us)
        out["single_instance_us_per_iter"] = round(best, 3)
        out["single_instance_note"] = "B = 1 runs in the team kernel (csrc/fused_team_kernel.hpp: four waves per instance); single_instance_one_wave_us_per_iter is the sweep kernel (one wave per instance) on the same transient"
        prev = os.environ.get("CADNIP_F2_TEAM")
        os.environ["CADNIP_F2_TEAM"] = "0"
        try:
            u0, conv, _ = sim.dc(abstol=1e-9, mode="tranop", fused=True)
            sim.h.set_spec(mode="tran")
            _, _, stats = sim.h.tran_run(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, RELTOL, breaks=breaks, save_t=np.array([bm.DFF_TSPAN[1]]),
                                         obs=[st.index_of("Q")], fused=int(args.fused) or 2, newton_mode=args.newton_mode if int(args.fused) else 1)
            out["single_instance_one_wave_us_per_iter"] = round(1e6 * stats["wall_seconds"] / max(stats["newton_iters"], 1), 3)
        finally:
            if prev is None:
                del os.environ["CADNIP_F2_TEAM"]
            else:
                os.environ["CADNIP_F2_TEAM"] = prev
        # callback sequence of one Newton iteration as a host integrator drives it: host pointers in, host pointers out
        h = sim.h
        rng = np.random.default_rng(0)
        du, gam, rhs = rng.random((1, st.n)), np.array([1e9]), rng.random((1, st.n))
        for rep in range(2):
            t0 = time.perf_counter()
            n_seq = 200
            for k in range(n_seq):
                h.rebuild(u0, 1e-8)
                h.residual(du, u0)
                h.jacobian(gam, readback=False)
                h.factor()
                h.solve(rhs)
            t1 = time.perf_counter()
        out["callback_us_per_iter"] = round(1e6 * (t1 - t0) / n_seq, 2)
        # the same iteration through cadnip_newton_step (one call, HIP graph replay) and cadnip_newton_step_fused (one kernel): raw ctypes calls on
        # preallocated buffers -- what a compiled host (Julia's ccall) pays; the Python wrapper's array handling costs ~13 us on top of each
        import ctypes as C
        dp = lambda a_: a_.ctypes.data_as(C.POINTER(C.c_double))
        uu, dd, gg, tt = np.ascontiguousarray(u0, dtype=np.float64), np.ascontiguousarray(du), np.ascontiguousarray(gam), np.array([1e-8])
        delta, nrm = np.empty_like(uu), np.empty(1)
        full = (h.h, dp(uu), dp(dd), dp(gg), dp(tt), C.c_int32(1), dp(delta), dp(nrm), None)
        kept = (h.h, dp(uu), dp(dd), None, None, C.c_int32(0), dp(delta), dp(nrm), None)
        for key, fn, a_ in (("callback_one_call_us_per_iter", h.lib.cadnip_newton_step, full), ("callback_one_call_kept_factors_us_per_iter", h.lib.cadnip_newton_step, kept),
                            ("callback_one_call_fused_us_per_iter", h.lib.cadnip_newton_step_fused, full),
                            ("callback_one_call_fused_kept_factors_us_per_iter", h.lib.cadnip_newton_step_fused, kept)):
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                for k in range(n_seq):
                    rc = fn(*a_)
                t1 = time.perf_counter()
                best = (t1 - t0) if best is None else min(best, t1 - t0)
            out[key] = round(1e6 * best / n_seq, 2) if rc == 0 else None
        out["callback_note"] = ("B = 1, through ctypes with host pointers: callback_us_per_iter = cadnip_rebuild -> residual -> jacobian -> factor -> solve (five calls, five "
                                "synchronisations, Python wrapper); callback_one_call_* = cadnip_newton_step (the same kernels, one call, HIP graph; bit for bit the five calls); "
                                "callback_one_call_fused_* = cadnip_newton_step_fused (one kernel: the team kernel's STEP mode, agrees to rounding) -- the one-call figures are raw "
                                "C-ABI calls on preallocated buffers (a compiled host's cost)")
    finally:
        sim.close()
    return out


def cpu_baseline(args, circ, sim, pts, save_t):
    """The oracle's C++ port on the host cores, bounded sample of the same corner grid.  Sweep points are independent, so the
    CPU farm is one port per point on a thread pool (the ctypes calls release the GIL); ports are built untimed."""
    from concurrent.futures import ThreadPoolExecutor
    from tests.port_util import make_port, analyze_port
    from oracle import cpu_port
    from cadnip_jl_amd import benchmarks as bm
    from cadnip_jl_amd.structure import TYPE_ID, pack_params, expand_breakpoints
    st, B = sim.st, len(pts)
    atol = st.state_abstol(**ABSTOL)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    obs = [st.index_of("Q")]
    try:
        n_all = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n_all = os.cpu_count() or 1
    n_thr = max(1, min(16, n_all))
    n_sample = min(B, max(args.cpu_sample, 32 * n_thr, 8 * n_all))
    idx = list(range(0, B, max(1, B // n_sample)))[:n_sample]
    sample = [pts[i] for i in idx]
    u0_all, _, _ = sim.dc(abstol=1e-9, mode="tranop")
    packed = pack_params(st, circ, sim.params, sim.temps, B)
    pst0, port0 = make_port(circ, {"vdd": sample[0]["vdd"]}, sample[0]["temp"], "tranop")
    prog = analyze_port(pst0, port0, sim.vscale())
    port0.close()
    ports = []
    for i in idx:
        port = cpu_port.Port(st, [blk[i] for blk in packed], TYPE_ID)
        port.set_spec(mode="tranop", gmin=1e-12)
        port.set_lu(prog)
        ports.append((st, port))

    def one(k):
        pst, port = ports[k]
        port.set_spec(mode="tranop")
        u0c, ok, dit = port.dc(abstol=1e-9)
        port.set_spec(mode="tran")
        _, _, rst, _ = port.tran(u0c if ok else u0_all[idx[k]], bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, RELTOL, breaks=breaks,
                                 save_t=save_t, obs=obs, err_mask=pst.differential_mask(), use_pcnr=False,
                                 newton_mode=args.newton_mode if int(args.fused) or not args.newton_mode else 2)   # (port mode 2 = the per-op GPU path's mode 1)
        return rst["newton_iters"] + dit

    n1 = max(8, len(sample) // 8)                   # single-thread leg on a slice, multi-thread leg on the whole sample
    tc0 = time.perf_counter()
    it1 = sum(one(k) for k in range(n1))
    t1 = time.perf_counter() - tc0
    tc0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=n_thr) as ex:
        itn = sum(ex.map(one, range(len(sample))))
    tn = time.perf_counter() - tc0
    # the whole host: one thread per CPU this process may run on (the figure the GPU has to be compared with; the 16-thread leg stays for
    # continuity with earlier rounds)
    ta, ita = tn, itn
    if n_all > n_thr:
        tc0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=n_all) as ex:
            ita = sum(ex.map(one, range(len(sample))))
        ta = time.perf_counter() - tc0
    for _, port in ports:
        port.close()
    # `value` is the best MEASURED farm.  A one-GPU box of this pool gives the job a share of the host (16 CPUs; its cgroup does not show in the
    # affinity mask, so the all-CPU leg usually measures the same 16), hence the estimate for the whole host beside it: one core's rate
    # times the physical cores -- the figure a reader should hold the GPU against (sweep points are independent: a farm scales).
    best_v, best_c = (ita / ta, n_all) if ita / ta > itn / tn else (itn / tn, n_thr)
    phys = max(1, (os.cpu_count() or 2) // 2)
    return {"value": round(best_v, 1), "unit": "newton_iters/s", "cores": best_c, "kind": "port", "cpu_model": cpu_model(),
            "host_logical_cpus": os.cpu_count(), "value_16_threads": round(itn / tn, 1), "value_all_cpu_threads": round(ita / ta, 1), "all_cpu_threads": n_all,
            "whole_host_estimate": {"value": round(it1 / t1 * phys, 1), "physical_cores": phys, "how": "value_1_core x physical cores (SMT off): an upper bound for a farm of independent transients"},
            "sample": "%d of the %d corner points (same DFF transient, same tolerances), oracle/cpu_port.cpp -O3 -march=native, "
                      "one port per point on a thread pool of %d threads (`value_16_threads`) and of one thread per visible CPU (%d; `value_all_cpu_threads`), newton_mode %d like the GPU run; "
                      "the port takes its pivot order from the product's host symbolic phase" % (len(sample), B, n_thr, n_all, args.newton_mode),
            "seconds": round(tn + t1 + (ta if n_all > n_thr else 0.0), 2), "value_1_core": round(it1 / t1, 1), "us_per_iter_1_core": round(1e6 * t1 / max(it1, 1), 2)}


if __name__ == "__main__":
    main()
