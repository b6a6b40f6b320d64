"""Multi-GPU sharding of the solver path: independent simulation islands are assigned whole to ranks (one process per GPU);
constraints never couple bodies of different islands, so the data path needs no collective — only the end-of-frame
barrier and the max-over-ranks timing reduction go through torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def partition_islands(island_sizes: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time bin packing of islands (by constraint count) onto ranks. Deterministic."""
    order = sorted(range(len(island_sizes)), key=lambda i: (-island_sizes[i], i))
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += island_sizes[i]
    return [sorted(x) for x in out]


def rank_seed(base_seed: int, rank: int) -> int:
    """Weak-scaling benchmark: every rank generates its own islands from a distinct seed."""
    return base_seed + rank


def aggregate_throughput(dist, units_this_rank: int, elapsed_this_rank: float, device=None) -> float:
    """Whole-job units/s: total units over all ranks / max elapsed over ranks (the bench contract)."""
    import torch
    if dist is None or not dist.is_initialized():
        return units_this_rank / elapsed_this_rank
    t = torch.tensor([elapsed_this_rank], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_this_rank)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item())


def connected_components(body_count: int, constraint_bodies: np.ndarray) -> np.ndarray:
    """Island label per body from [n, 2] dynamic body index pairs (-1 for absent / kinematic slots). Union-find, path halving."""
    parent = np.arange(body_count, dtype=np.int64)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in constraint_bodies:
        if a < 0 or b < 0:
            continue
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return np.asarray([find(i) for i in range(body_count)], dtype=np.int64)
