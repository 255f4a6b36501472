"""bench.py's output contract, run in-process (the library and torch share one HIP runtime in either load order:
hip.py _share_hip_runtime, tests/test_gpu_coexist.py -- this module no longer has to run first)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
def test_bench_line_contract(capsys):
    """bench.py in-process on a small sweep: ONE JSON line with the contract fields, the roofline object of the dominant
    kernel (live HIP-event timing) and the CPU baseline of the oracle's port."""
    import json
    import torch
    assert torch.cuda.is_available()
    import bench
    bench.main(["--steps", "1", "--warmup", "0", "--instances", "64", "--cpu-sample", "32", "--total-instances", "64"])
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "newton_iters_per_sec" and d["unit"] == "newton_iters/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["instances_per_gpu"] == 64 and d["value"] > 1e6
    r = d["roofline"]
    # the fused kernel is bound by vector-instruction issue, not by HBM: frac is a utilisation, so it lies in (0, 1]
    assert r["bound"] == "valu_issue" and r["avg_launch_us"] > 0 and r["source"]
    assert r["frac"] is not None and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if "hbm" in r:
        assert r["hbm"]["bound"] == "hbm" and r["hbm"]["peak"] == 8000.0 and 0.0 < r["hbm"]["frac"] <= 1.0
    s = d["stamp_kernel"]
    # frac: the counter figure when a counter pass ran (HBM bytes moved / duration / peak), SURVEY 8d's algorithmic figure beside it
    assert s["bound"] == "hbm" and s["B"] == 8192 and 0.0 < s["frac"] <= 1.0 and abs(s["frac_algorithmic"] - s["achieved_GBps"] / 8000.0) < 1e-3
    assert abs(s["frac"] - (s["traffic_GBps"] if s.get("traffic") else s["achieved_GBps"]) / 8000.0) < 1e-3 and s["frac_source"]
    assert d["transients_per_s"] > 0 and 0.0 <= d["rejected_step_share"] < 0.5 and d["newton_iters_per_accepted_step"] > 1.0
    assert d["single_instance_one_wave_us_per_iter"] > d["single_instance_us_per_iter"]       # the team kernel is what a single transient gets
    assert d["strong_1024"]["instances_total"] == 64 and d["strong_1024"]["value"] > 0
    assert d["single_instance_us_per_iter"] > 0 and d["callback_us_per_iter"] > 0
    assert d["config"]["newton_mode"] == 1 and d["full_newton"]["newton_iters_per_step"] > d["config"]["newton_iters_per_step"]
    p = d["psp103_ring"]
    assert "error" not in p and p["B1"]["failed"] == 0 and p["B256"]["failed"] == 0 and p["B256"]["us_per_instance_iter"] < p["B1"]["us_per_instance_iter"] < 1376.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and "sample" in c

