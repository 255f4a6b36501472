"""Condense one profiling session into profiles/<tag>_pmc_summary.json: the counter figures of the fused kernel that bench.py
collected live (its rocprofv3 --pmc child passes) plus the LDS-unit pass; bench.py falls back to this file when it cannot run
rocprofv3 itself.

usage: python tools/make_pmc_summary.py <tag> <bench line .json> [<lds pmc.json from tools/pmc_summary.py>]
"""
import json
import sys


def main():
    tag, bench_fn = sys.argv[1:3]
    bench = json.load(open(bench_fn))
    r = bench["roofline"]
    assert r["source"].startswith("live"), "the bench line carries no live counter pass"
    out = {
        "note": "rocprofv3 --pmc <counters> --kernel-trace child passes of `python3 bench.py` (one pass per counter set: SQ_*, FETCH_SIZE, "
                "WRITE_SIZE; the HBM passes with the 1 GiB fp64 calibration copy); %d instances, fused kernel, 1024 Newton rounds per launch. "
                "FETCH_SIZE / WRITE_SIZE are in KiB and corrected with the calibration factors below (gfx950: FETCH_SIZE reads half the true "
                "bytes); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles (x 4 = cycles)." % bench["config"]["instances_per_gpu"],
        "bench_value_newton_iters_per_s": bench["value"],
        "fused2_newton": {
            "instances": bench["config"]["instances_per_gpu"], "rounds_per_launch": 1024,
            "valu_busy_simd_cycles_per_ns": r["achieved"], "valu_busy_frac": r["frac"], "peak_simd_cycles_per_ns": r["peak"],
            "hbm_bytes_per_launch": r.get("traffic"), "hbm": r.get("hbm"),
            "wave_time_shares": r["wave_time_shares"], "per_newton_round": r["per_newton_round"], "valu_insts_per_s_G": r["valu_insts_per_s"]},
        "stamp_kernel": bench.get("stamp_kernel"),
    }
    if len(sys.argv) > 3:
        d = json.load(open(sys.argv[3]))
        K = next(k for k in d if "k_fused2<" in k and ", false, " in k)
        l = d[K]
        out["fused2_newton"]["lds_unit"] = {
            "note": "separate pass: SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_LDS_ADDR_CONFLICT over SQ_BUSY_CU_CYCLES",
            "lds_active_share_of_cu_busy": l["SQ_LDS_IDX_ACTIVE"]["total"] / l["SQ_BUSY_CU_CYCLES"]["total"],
            "bank_conflict_share_of_lds_active": l["SQ_LDS_BANK_CONFLICT"]["total"] / l["SQ_LDS_IDX_ACTIVE"]["total"],
            "same_address_conflict_share_of_lds_active": l["SQ_LDS_ADDR_CONFLICT"]["total"] / l["SQ_LDS_IDX_ACTIVE"]["total"]}
    json.dump(out, open("%s_pmc_summary.json" % (tag if "/" in tag else "profiles/" + tag), "w"), indent=1)
    print(json.dumps(out["fused2_newton"], indent=1))


if __name__ == "__main__":
    main()
