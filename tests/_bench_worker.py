"""Worker for tests/test_multiprocess_cpu.py::test_bench_control_flow_world_2: bench.run() -- the benchmark's whole control
flow between process-group set-up and printing -- on gloo CPU ranks with a stub in place of the GPU simulator.  What it
guards: every rank takes part in every collective (a rank-0-only profiled step must not gather: that deadlocked N > 1),
both workloads (weak, strong) run on every rank, and rank 0 alone assembles the line."""
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cadnip_jl_amd as cj  # noqa: E402
import bench  # noqa: E402

CALLS = []


class StubHandle:
    def __init__(self, st, B):
        self.st, self.B, self._prof = st, B, False

    def set_spec(self, **kw):
        pass

    def tran_run(self, t0, t1, atol, reltol, breaks=(), save_t=(), obs=None, fused=0, newton_mode=0):
        CALLS.append("tran")
        time.sleep(0.01)
        out = np.full((self.B, len(save_t), len(obs)), float(dist.get_rank()))
        return out, np.zeros((self.B, 4), dtype=np.int64), {"newton_iters": 100 * self.B, "n_failed": 0, "launches": 3,
                                                             "steps_accepted": 50 * self.B, "steps_rejected": 2 * self.B, "wall_seconds": 0.01}

    def profile(self, on):
        self._prof = on

    def profile_read(self):
        return {"fused2_newton": (3.0, 3), "fused2_dc": (0.1, 1)}

    def lu_stats(self):
        return {"nnz_lu": self.st.nnz + 1}


class StubSim:
    def __init__(self, circ, pts, device):
        self.st = cj.discover(circ, {"vdd": 5.0})
        self.B = len(pts)
        self.h = StubHandle(self.st, self.B)

    def analyze(self):
        pass

    def dc(self, abstol=1e-9, mode="tranop", fused=False):
        return np.zeros((self.B, self.st.n)), np.ones(self.B, dtype=bool), {"newton_iters": 10 * self.B}

    def close(self):
        pass


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allreduce(x, op):
        t = torch.tensor([x], dtype=torch.float64 if isinstance(x, float) else torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return float(t.item()) if isinstance(x, float) else int(t.item())

    args = types.SimpleNamespace(steps=2, warmup=1, instances=64, total_instances=96, fused=2, no_cpu_baseline=True, no_live_pmc=True,
                                 no_extras=True, cpu_sample=8, newton_mode=1)
    res = bench.run(args, rank, world, rank, dist, None, sync=lambda: None, reduce_max=lambda x: allreduce(x, "max"),
                    reduce_sum=lambda x: allreduce(x, "sum"), simulator=StubSim)
    dist.barrier()
    dist.destroy_process_group()
    # weak: (1 warmup + 2 timed) steps; strong: (1 + 2); rank 0: + 1 profiled step and 2 full-Newton comparison steps, none of them gathering
    assert CALLS.count("tran") == 6 + (3 if rank == 0 else 0), CALLS
    if rank == 0:
        assert res["n_gpus"] == world and res["config"]["instances_total"] == 64 * world
        assert res["config"]["newton_iters_per_step"] == 110 * 64 * world                  # SUM over ranks
        assert res["strong_1024"]["instances_total"] == 96 and res["strong_1024"]["instances_per_gpu"] == 96 // world
        assert res["roofline"]["kernel"] == "fused2_newton" and res["roofline"]["bound"] == "valu_issue"
        print("BENCH_FLOW_OK " + json.dumps({"value": res["value"], "strong": res["strong_1024"]["value"]}))
    else:
        assert res is None


if __name__ == "__main__":
    main()
