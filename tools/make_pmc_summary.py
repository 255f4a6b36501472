"""Condense the rocprofv3 passes of one profiling session into profiles/<tag>_pmc_summary.json.

usage: python tools/make_pmc_summary.py <tag> <bench.json> <pmc.json from tools/pmc_summary.py> [<lds pmc.json>]
"""
import json, sys

CAL = 'cadnip::k_calib_copy_f64(double const*, double*, long)'


def main():
    tag, bench_fn, pmc_fn = sys.argv[1:4]
    d = json.load(open(pmc_fn))
    bench = json.load(open(bench_fn))
    # the transient instantiation: k_fused2<WPB, DC=false, DIRECT>
    K = next(k for k in d if 'k_fused2<' in k and ', false, ' in k)
    f, cal = d[K], d[CAL]
    fc = 1024 * 1024 / cal['FETCH_SIZE']['per_call']
    wc = 1024 * 1024 / cal['WRITE_SIZE']['per_call']
    calls = f['FETCH_SIZE']['calls']
    hbm_total = (f['FETCH_SIZE']['total'] * fc + f['WRITE_SIZE']['total'] * wc) * 1024
    iters = 2 * bench['config']['newton_iters_per_step']      # every profiled command runs two transients (timed step + roofline step)
    wc_tot = f['SQ_WAVE_CYCLES']['total']
    out = {
        "note": "rocprofv3 --pmc <counters> --kernel-trace, one pass per counter set; command: python3 bench.py --steps 1 --warmup 0 "
                "--no-cpu-baseline [--calib-copy 1024 for the HBM passes] (%d instances, fused kernel, 1024 Newton rounds per launch). "
                "Units: KB as reported. k_calib_copy_f64 streams 1 GiB in + 1 GiB out per call: FETCH_SIZE reads 1/2 of the true bytes "
                "on gfx950 (correction x2, MI355X_MICROARCH.md HBM section), WRITE_SIZE is exact. SQ_WAVE_CYCLES / SQ_WAIT_* / "
                "SQ_ACTIVE_INST_* count quad-cycles." % bench['config']['instances_per_gpu'],
        "calibration": {"fetch_correction": round(fc, 4), "write_correction": round(wc, 4)},
        "kernels": {K: f, CAL: cal},
        "fused2_newton": {
            "instances": bench['config']['instances_per_gpu'], "rounds_per_launch": 1024, "launches_profiled": calls,
            "hbm_bytes_per_launch": int(hbm_total / calls), "hbm_bytes_total": int(hbm_total),
            "wave_time_shares": {"waiting_on_waitcnt": f['SQ_WAIT_ANY']['total'] / wc_tot, "issue_stalled": f['SQ_WAIT_INST_ANY']['total'] / wc_tot,
                                 "issuing": f['SQ_ACTIVE_INST_ANY']['total'] / wc_tot, "issuing_valu": f['SQ_ACTIVE_INST_VALU']['total'] / wc_tot},
            "per_newton_round": {"hbm_bytes": hbm_total / iters, "valu_insts": f['SQ_INSTS_VALU']['total'] / iters,
                                 "salu_insts": f['SQ_INSTS_SALU']['total'] / iters, "lds_insts": f['SQ_INSTS_LDS']['total'] / iters,
                                 "wave_cycles": 4 * wc_tot / iters}}}
    if len(sys.argv) > 4:
        l = json.load(open(sys.argv[4]))[K]
        out["fused2_newton"]["lds_unit"] = {
            "note": "separate pass: SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_LDS_ADDR_CONFLICT over SQ_BUSY_CU_CYCLES",
            "lds_active_share_of_cu_busy": l['SQ_LDS_IDX_ACTIVE']['total'] / l['SQ_BUSY_CU_CYCLES']['total'],
            "bank_conflict_share_of_lds_active": l['SQ_LDS_BANK_CONFLICT']['total'] / l['SQ_LDS_IDX_ACTIVE']['total'],
            "same_address_conflict_share_of_lds_active": l['SQ_LDS_ADDR_CONFLICT']['total'] / l['SQ_LDS_IDX_ACTIVE']['total']}
    json.dump(out, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1)
    print(json.dumps(out["fused2_newton"], indent=1))


if __name__ == "__main__":
    main()
