#!/usr/bin/env python3
"""bench.py -- Newton iterations/sec on the synthetic gf180 DFF transient (BASELINE.json metric).

One "step" = one complete tran! of the per-GPU sweep batch: CedarTranOp-style DC initialisation
(:tranop, PCNR Newton) + the 0-700 ns transient of every resident sweep instance, results gathered.
Per-GPU work is fixed (weak scaling): each rank integrates ``--instances`` corner points (default 4096:
BASELINE.json config 4's 32-value Vdd axis x 128 temperatures, so that the 2048 instances a GPU holds at a time
are followed by a second generation from the in-kernel queue); with N ranks the temp axis is refined N-fold over
the same range and split into contiguous blocks, so N GPUs process N x 4096 independent
transients.  The only collective is the final gather of the result
blocks (RCCL all_gather), inside the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : dominant kernel, algorithmic bytes per launch / its measured average
                 launch duration (HIP events on the launching stream), vs 8 TB/s HBM3E peak
  cpu_baseline : the oracle's C++ port (oracle/cpu_port.cpp) on a bounded sample of the same workload, one port per
                 corner point on 16 host threads; `value_1_core` is the single-thread rate (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ABSTOL = dict(vntol=1e-6, iabstol=1e-9, chgtol=1e-6)
RELTOL = 1e-4


PMC_SUMMARY = "r01h_fused_B4096_pmc_summary.json"   # committed rocprofv3 --pmc passes of this kernel at this batch size


def algorithmic_bytes(st, B, nnz_lu, rounds=8):
    """Per-launch algorithmic bytes of each hot-path kernel for a batch of B instances
    (SURVEY.md section 8d / DESIGN.md section 5: every array counted once per required read or write)."""
    out = {}
    for blk in st.blocks:
        per_dev = 8 * blk.n_par + 4 * blk.nodes.shape[0] + 8 * (blk.n_g + blk.n_c + blk.n_b)
        if blk.type in ("D", "MOS1"):
            per_dev += 8 * (1 if blk.type == "D" else 4)            # limit_w
        out["stamp_" + {"R": "resistor", "C": "capacitor", "L": "inductor", "V": "vsource", "I": "isource", "E": "vcvs",
                        "G": "vccs", "H": "ccvs", "F": "cccs", "D": "diode", "DCAP": "diodecap", "SMOS": "simplemos",
                        "MOS1": "mos1"}[blk.type]] = B * (blk.count * per_dev + 8 * st.n)
    ns = st.ns_g + st.ns_c + st.ns_b
    n_coo = st.n_coo_g + st.n_coo_c + st.n_coo_b
    out["assemble"] = B * (8 * n_coo + 8 * (2 * st.nnz + st.n)) + 4 * (n_coo + 2 * st.nnz + st.n)
    out["residual"] = B * (2 * 8 * st.nnz + 8 * 4 * st.n) + 4 * (st.nnz + st.n + 1)
    out["lu_factor_solve"] = B * (8 * 2 * st.nnz + 8 * nnz_lu + 8 * 2 * st.n)
    out["tran_update"] = B * 8 * 8 * st.n
    # fused kernels: one launch = `rounds` Newton iterations of every instance; B_iter = sum of the per-op rows
    b_iter = sum(v for k, v in out.items() if k.startswith("stamp_")) + out["assemble"] + out["residual"] + out["lu_factor_solve"] + out["tran_update"]
    out["fused_newton"] = rounds * b_iter
    out["fused2_newton"] = rounds * b_iter
    out["B_iter_per_instance"] = b_iter // B
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=4096, help="sweep instances per GPU (32 Vdd x instances/32 temps); 2048 are resident at a time, the rest queue in-kernel")
    ap.add_argument("--cpu-sample", type=int, default=512, help="corner points timed on the host for cpu_baseline")
    ap.add_argument("--fused", type=int, default=int(os.environ.get("CADNIP_FUSED", "2")),
                    help="0 = one kernel per op (the kernels behind the callback ABI), non-zero = fused Newton kernel (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calib-copy", type=int, default=0,
                    help="also run the 8 B/lane fp64 calibration copy of this many MiB (for rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE runs)")
    args = ap.parse_args(argv)

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

    import cadnip_jl_amd  # noqa: F401
    from cadnip_jl_amd import api, benchmarks as bm, sweep_shard
    from cadnip_jl_amd.structure import expand_breakpoints

    circ = bm.dff_circuit()
    pts, _ = sweep_shard.rank_points(args.instances, rank, world)
    B = len(pts)
    sim = api.BatchSimulator(api.MNACircuit(circ, {"vdd": 5.0}), pts, device=local_rank)
    st = sim.st
    sim.analyze()
    atol = st.state_abstol(**ABSTOL)
    breaks = expand_breakpoints(st.breakpoints, bm.DFF_TSPAN)
    save_t = np.linspace(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], 71)
    obs = [st.index_of("Q")]
    dev = torch.device("cuda", local_rank)

    def one_step():
        u0, conv, dcs = sim.dc(abstol=1e-9, mode="tranop", fused=bool(args.fused))
        if not np.all(conv):
            raise RuntimeError("DC initialisation failed on rank %d" % rank)
        sim.h.set_spec(mode="tran")
        out, per, stats = sim.h.tran_run(bm.DFF_TSPAN[0], bm.DFF_TSPAN[1], atol, RELTOL, breaks=breaks, save_t=save_t, obs=obs,
                                         fused=int(args.fused))
        if stats["n_failed"]:
            raise RuntimeError("%d transient(s) failed on rank %d" % (stats["n_failed"], rank))
        if world > 1:   # final gather of the result blocks over RCCL / xGMI
            sweep_shard.gather_blocks(out, world, dev)
        return stats["newton_iters"] + dcs["newton_iters"], out, stats

    if args.calib_copy:
        sim.h.debug_copy(args.calib_copy * (1 << 20) // 8, 4)
    for _ in range(args.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    last = None
    for _ in range(args.steps):
        n_it, out, last = one_step()
        iters += n_it
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        it = torch.tensor([iters], device=dev, dtype=torch.int64)
        dist.all_reduce(it, op=dist.ReduceOp.SUM)
        iters = int(it.item())

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: one extra, event-timed step (not part of `value`) -------
        sim.h.profile(True)
        _, _, pstats = one_step()
        prof = sim.h.profile_read()
        sim.h.profile(False)
        ab = algorithmic_bytes(st, B, sim.h.lu_stats()["nnz_lu"])
        dom = max(prof.items(), key=lambda kv: kv[1][0])
        kernels = {k: {"ms_total": round(v[0], 3), "calls": int(v[1]), "avg_us": round(1e3 * v[0] / max(v[1], 1), 3)} for k, v in prof.items()}
        name = dom[0]
        avg_s = dom[1][0] / max(dom[1][1], 1) * 1e-3
        roof = {"bound": "hbm", "kernel": name, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "avg_launch_us": round(avg_s * 1e6, 3), "kernels": kernels}
        roof["B_iter_per_instance"] = int(ab["B_iter_per_instance"])
        if name in ab:
            per_launch = ab[name]
            if name.startswith("fused"):
                # a fused launch runs up to 8 Newton rounds of every instance still active: count the iterations actually done
                per_launch = ab["B_iter_per_instance"] * pstats["newton_iters"] / max(dom[1][1], 1)
                roof["newton_iters_per_launch"] = round(pstats["newton_iters"] / max(dom[1][1], 1), 1)
            if name.startswith("fused"):
                roof["note"] = ("algorithmic bytes = SURVEY.md 8d's B_iter (slot writes, G/C/J, LU factors counted as memory traffic) x Newton "
                                "iterations executed; the fused kernel keeps all of that in LDS and registers, so frac can exceed 1 and "
                                "`traffic` (PMC HBM bytes) is far below it -- the kernel is VALU-issue / LDS-latency bound, see wave_time_shares")
            roof["algorithmic_bytes_per_launch"] = int(per_launch)
            roof["achieved"] = round(per_launch / avg_s / 1e9, 3)
            roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 5)
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950
        # correction calibrated with k_calib_copy_f64; profiles/*_pmc_summary.json) -- only when it is the same kernel / batch
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_SUMMARY)))
            if name in pmc and pmc[name]["instances"] == B:
                roof["traffic"] = pmc[name]["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/" + PMC_SUMMARY
                roof["wave_time_shares"] = {k: round(v, 3) for k, v in pmc[name].get("wave_time_shares", {}).items()}
        except (OSError, ValueError, KeyError):
            pass
        if "stamp_mos1" in prof and name != "stamp_mos1":
            s_avg = prof["stamp_mos1"][0] / max(prof["stamp_mos1"][1], 1) * 1e-3
            roof["stamp_mos1_GBps"] = round(ab["stamp_mos1"] / s_avg / 1e9, 3)
        # ---- CPU baseline: the oracle's C++ port on the host cores, bounded sample of the same corner grid -----------
        # Sweep points are independent, so the CPU farm is one port per point on a thread pool (the ctypes calls release the
        # GIL: the threads run the compiled port concurrently).  Ports are built untimed; the timed region is DC + transient.
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from concurrent.futures import ThreadPoolExecutor
            from tests.port_util import make_port, analyze_port
            from oracle import cpu_port
            from cadnip_jl_amd.structure import TYPE_ID, pack_params
            n_thr = max(1, min(16, os.cpu_count() or 1))
            n_sample = min(B, max(args.cpu_sample, 32 * n_thr))
            idx = list(range(0, B, max(1, B // n_sample)))[:n_sample]
            sample = [pts[i] for i in idx]
            u0_all, _, _ = sim.dc(abstol=1e-9, mode="tranop")
            # one port per sampled point from the batch's own parameter blocks; the symbolic LU is shared (one structure)
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
                                         save_t=save_t, obs=obs, err_mask=pst.differential_mask(), use_pcnr=False)
                return rst["newton_iters"] + dit

            n1 = max(8, len(sample) // 8)                   # single-thread leg on a slice, multi-thread leg on the whole sample
            tc0 = time.perf_counter()
            it1 = sum(one(k) for k in range(n1))
            t1 = time.perf_counter() - tc0
            tc0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=n_thr) as ex:
                itn = sum(ex.map(one, range(len(sample))))
            tn = time.perf_counter() - tc0
            for _, port in ports:
                port.close()
            cpu = {"value": round(itn / tn, 1), "unit": "newton_iters/s", "cores": n_thr, "kind": "port",
                   "sample": "%d of the %d corner points (same DFF transient, same tolerances), oracle/cpu_port.cpp -O3 -march=native, "
                             "one port per point on %d host threads" % (len(sample), B, n_thr),
                   "seconds": round(tn + t1, 2), "value_1_core": round(it1 / t1, 1)}
        result = {
            "metric": "newton_iters_per_sec", "value": round(iters / elapsed, 1), "unit": "newton_iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "gf180 DFF transient 0-700 ns (synthetic sp_mos1 level-1 card, 30 MOSFETs, n=%d, nnz=%d), "
                                   "%d Vdd x temp corner instances per GPU (state resident in HBM; 8 per CU in flight, the rest handed out by the kernel's instance queue); step = DC init + full transient of the batch"
                                   % (st.n, st.nnz, B),
                       "instances_per_gpu": B, "instances_total": B * world, "abstol": ABSTOL, "reltol": RELTOL,
                       "fused": int(args.fused), "newton_iters_per_step": int(iters // max(args.steps, 1)),
                       "launches_last_step": int(last["launches"]), "accepted_last_step": int(last["steps_accepted"]),
                       "rejected_last_step": int(last["steps_rejected"])},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(result))
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
