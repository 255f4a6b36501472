"""Worker for tests/test_multiprocess_cpu.py: one process per rank, gloo backend, no GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cadnip_jl_amd  # noqa: E402,F401
from cadnip_jl_amd import sweep_shard  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_per = int(sys.argv[1])
    pts, n_total = sweep_shard.rank_points(n_per, rank, world)
    assert len(pts) == n_per and n_total == n_per * world
    # a result block whose rows encode the point itself: after the gather, row g must describe global point g
    block = np.array([[p["vdd"], p["temp"], float(rank)] for p in pts])
    full = sweep_shard.gather_blocks(block, world)
    ref = sweep_shard.corner_points(32 if n_per >= 32 else n_per, max(1, n_per // (32 if n_per >= 32 else n_per)) * world)
    assert full.shape == (n_total, 3)
    assert np.allclose(full[:, 0], [p["vdd"] for p in ref]) and np.allclose(full[:, 1], [p["temp"] for p in ref])
    assert np.array_equal(full[:, 2], np.repeat(np.arange(world), n_per).astype(float))
    # the two reductions bench.py uses: MAX over ranks of the elapsed time, SUM of the iteration counts
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    it = torch.tensor([100 * (rank + 1)], dtype=torch.int64)
    dist.all_reduce(it, op=dist.ReduceOp.SUM)
    assert t.item() == float(world) and it.item() == 100 * world * (world + 1) // 2
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("GLOO_OK", n_total)


if __name__ == "__main__":
    main()
