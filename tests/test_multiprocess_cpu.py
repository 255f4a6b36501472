"""N > 1 path on CPU: world_size-2 gloo run of the sweep-farm partition + final gather used by bench.py."""
import os
import socket
import subprocess
import sys

import numpy as np

from cadnip_jl_amd import sweep_shard

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_block_partition_is_contiguous_and_complete():
    for n, w in ((1024, 8), (1000, 8), (7, 3), (5, 8)):
        got = []
        for r in range(w):
            lo, hi = sweep_shard.block_range(n, r, w)
            got += list(range(lo, hi))
        assert got == list(range(n))
    pts = sweep_shard.corner_points(32, 32)
    assert len(pts) == 1024 and pts[0] == {"vdd": 4.5, "temp": -40.0} and pts[1]["temp"] == -40.0 and pts[1]["vdd"] > 4.5
    assert pts[32]["vdd"] == 4.5 and pts[32]["temp"] > -40.0        # Vdd is the fast axis


def test_world_size_2_gloo_gather():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), "64"], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK 128" in outs[0]


def test_strong_partition_and_unequal_blocks():
    """Strong scaling (config 4 as worded): 1024 points over 8 GPUs = 128 per GPU, contiguous in ProductSweep order."""
    allp = sweep_shard.corner_points(32, 32)
    got = []
    for r in range(8):
        pts, n = sweep_shard.strong_points(1024, r, 8)
        assert n == 1024 and len(pts) == 128
        got += pts
    assert got == allp
    pts, n = sweep_shard.strong_points(96, 1, 2)
    assert n == 96 and len(pts) == 48 and pts[0] == sweep_shard.corner_points(32, 3)[48]


def test_bench_control_flow_world_2():
    """bench.run on two gloo ranks with a stub simulator: no rank-0-only collective (the N > 1 deadlock of round 1), weak and
    strong workloads on every rank, SUM / MAX reductions, rank 0 assembles the line."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_bench_worker.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "BENCH_FLOW_OK" in outs[0]
