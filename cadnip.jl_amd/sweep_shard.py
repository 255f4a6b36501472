"""Sweep farm: static partition of independent sweep points across the GPUs of one node and the
single collective of the path -- the final gather of the fixed-size result blocks.

The reference runs a CircuitSweep as a serial loop over altered circuits
(/root/reference/src/sweeps.jl:692-707); the points never exchange state, so the farm needs no
data-path collective: each rank (one process per GPU) integrates a contiguous block of the
ProductSweep linear index and the blocks are all-gathered once at the end
(RCCL over xGMI on the GPU box; gloo in the CPU tests).
"""
import numpy as np


def corner_points(n_vdd, n_temp, vdd=(4.5, 5.5), temp=(-40.0, 125.0)):
    """Global Vdd x temp grid in ProductSweep order (Vdd fastest, sweeps.jl:272)."""
    vdds = np.linspace(vdd[0], vdd[1], n_vdd) if n_vdd > 1 else np.array([0.5 * (vdd[0] + vdd[1])])
    temps = np.linspace(temp[0], temp[1], n_temp) if n_temp > 1 else np.array([27.0])
    return [{"vdd": float(v), "temp": float(t)} for t in temps for v in vdds]


def block_range(n_total, rank, world):
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_points(n_per_gpu, rank, world):
    """Weak scaling: every rank owns ``n_per_gpu`` points; the global grid keeps 32 Vdd values and refines
    the temperature axis to (n_per_gpu / 32) * world values over the same range."""
    n_vdd = 32 if n_per_gpu >= 32 else n_per_gpu
    n_temp = max(1, n_per_gpu // n_vdd) * world
    pts = corner_points(n_vdd, n_temp)
    lo, hi = block_range(len(pts), rank, world)
    return pts[lo:hi], len(pts)


def strong_points(n_total, rank, world):
    """Strong scaling (BASELINE.json config 4 as worded): ONE global grid of ``n_total`` points -- 32 Vdd
    values x n_total / 32 temperatures in ProductSweep order -- split into contiguous blocks, 1024 / 8 = 128 points per GPU."""
    n_vdd = 32 if n_total >= 32 else n_total
    pts = corner_points(n_vdd, max(1, n_total // n_vdd))
    lo, hi = block_range(len(pts), rank, world)
    return pts[lo:hi], len(pts)


def gather_blocks(local_block, world, device=None):
    """All-gather the per-rank result blocks [B_local, ...] into the global [B_total, ...] array (rank order == sweep
    order).  Block sizes may differ by one row (block_range): blocks travel padded to the largest and are trimmed again.
    Works on whatever backend torch.distributed was initialised with."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_block
    local_block = np.ascontiguousarray(local_block)
    rows = torch.tensor([local_block.shape[0]], dtype=torch.int64)
    if device is not None:
        rows = rows.to(device)
    all_rows = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(all_rows, rows)
    all_rows = [int(r.item()) for r in all_rows]
    n_max = max(all_rows)
    if local_block.shape[0] < n_max:
        pad = np.zeros((n_max - local_block.shape[0],) + local_block.shape[1:], dtype=local_block.dtype)
        local_block = np.concatenate([local_block, pad], axis=0)
    t = torch.from_numpy(local_block)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return torch.cat([p[:r] for p, r in zip(parts, all_rows)], dim=0).cpu().numpy()
