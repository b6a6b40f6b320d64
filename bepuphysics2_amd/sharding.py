"""Multi-GPU sharding of the solver path: independent simulation islands are assigned whole to ranks (one process per GPU);
constraints never couple bodies of different islands, so the data path needs no collective — only the end-of-frame
barrier and the max-over-ranks timing reduction go through torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from dataclasses import dataclass
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
    """Island label per body (the smallest body index of its island) from [n, 2] dynamic body index pairs (-1 for absent / kinematic slots)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components as cc
    pairs = np.asarray(constraint_bodies, dtype=np.int64).reshape(-1, 2)
    pairs = pairs[(pairs[:, 0] >= 0) & (pairs[:, 1] >= 0)]
    graph = coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(body_count, body_count))
    _, labels = cc(graph, directed=False)
    first = np.full(labels.max() + 1 if body_count else 0, body_count, dtype=np.int64)
    np.minimum.at(first, labels, np.arange(body_count, dtype=np.int64))
    return first[labels]


KINEMATIC_BIT = 1 << 30
REF_MASK = 0x3FFFFFFF


@dataclass
class IslandShare:
    """One rank's part of a scene made of independent islands (BASELINE.json configs[3]): the bodies of its islands, private read-only copies of the
    kinematic bodies its constraints reference, its part of the unconstrained bodies, and the constraints of its islands IN THEIR ORIGINAL BATCHES
    and in their original relative order inside every type batch — so the order in which constraints are applied to any body is the single-process
    order, and the union of the shares' results is the single-process result bit for bit."""
    scene: object                 # bepuphysics2_amd.scene.Scene holding the share
    body_global: np.ndarray       # int64 [local bodies] -> body index in the whole scene
    owned: np.ndarray             # bool [local bodies]: written back by merge (kinematic copies are owned by exactly one rank)
    constraint_rows: list         # per (batch, type batch) of the WHOLE scene: indices of the constraints this share holds


def island_labels(scene) -> np.ndarray:
    """Island label per body: connected components through DYNAMIC references (kinematic bodies never connect islands, Solver.cs:1058-1078)."""
    pairs = []
    for batch in scene.batches:
        for tb in batch:
            r = tb.refs_lanes(scene.bundle_width).astype(np.int64)
            r = np.where((r & KINEMATIC_BIT) != 0, -1, r)
            for a in range(r.shape[1]):          # a constraint with n bodies connects all of them: chain them pairwise
                for b in range(a + 1, r.shape[1]):
                    pairs.append(r[:, [a, b]])
            if r.shape[1] == 1:
                pairs.append(np.stack([r[:, 0], r[:, 0]], axis=1))
    allpairs = np.concatenate(pairs) if pairs else np.zeros((0, 2), np.int64)
    return connected_components(scene.body_count, allpairs)


def split_scene_by_islands(scene, world_size: int, only_rank: int = -1) -> List[IslandShare]:
    """Partition a scene of independent islands onto `world_size` ranks (greedy bin packing by constraint count, partition_islands).
    only_rank >= 0 builds that rank's share alone (what a rank of a multi-process job needs); the list then holds one share."""
    from .scene import Scene, TypeBatchData, from_aosoa, to_aosoa, TYPE_TABLE
    w = scene.bundle_width
    labels = island_labels(scene)
    lanes = [[(tb, tb.refs_lanes(w).astype(np.int64)) for tb in batch] for batch in scene.batches]
    # island of a constraint = island of its first dynamic body (all its dynamic bodies share it by construction)
    constrained = np.zeros(scene.body_count, dtype=bool)
    con_island = []
    for batch in lanes:
        row = []
        for tb, r in batch:
            dyn = (r & KINEMATIC_BIT) == 0
            first_dyn = np.argmax(dyn, axis=1)
            has_dyn = dyn.any(axis=1)
            idx = r[np.arange(len(r)), first_dyn] & REF_MASK
            isl = np.where(has_dyn, labels[idx], -1)     # a constraint between kinematic bodies only belongs to nobody in particular: rank 0 takes it
            row.append(isl)
            constrained[(r & REF_MASK).ravel()] = True
        con_island.append(row)
    ids, counts = np.unique(np.concatenate([x for row in con_island for x in row]) if any(len(r) for r in con_island) else np.zeros(0, np.int64), return_counts=True)
    keep = ids >= 0
    ids, counts = ids[keep], counts[keep]
    parts = partition_islands(counts.tolist(), world_size)
    rank_of_island = np.full(scene.body_count + 1, -1, dtype=np.int64)
    for rank, part in enumerate(parts):
        rank_of_island[ids[part]] = rank
    kinematic = ~np.any(scene.bodies[:, 16:23] != 0, axis=1)
    shares = []
    free = np.nonzero(~constrained)[0]                       # unconstrained bodies: dealt round-robin, any rank can integrate them
    lowest_user = np.full(scene.body_count, world_size, dtype=np.int64)  # kinematic copies: only the lowest rank holding one writes it back
    for batch, isl_row in zip(lanes, con_island):
        for (tb, r), isl in zip(batch, isl_row):
            rk = np.where(isl >= 0, rank_of_island[isl], 0)
            np.minimum.at(lowest_user, (r & REF_MASK).ravel(), np.repeat(rk, r.shape[1]))
    for rank in (range(world_size) if only_rank < 0 else [only_rank]):
        rows, used = [], np.zeros(scene.body_count, dtype=bool)
        for batch, isl_row in zip(lanes, con_island):
            brow = []
            for (tb, r), isl in zip(batch, isl_row):
                mine = np.nonzero((rank_of_island[isl] == rank) | ((isl < 0) & (rank == 0)))[0]
                brow.append(mine)
                used[(r[mine] & REF_MASK).ravel()] = True
            rows.append(brow)
        used[free[rank::world_size]] = True
        body_global = np.nonzero(used)[0]
        local_of = np.full(scene.body_count, -1, dtype=np.int64)
        local_of[body_global] = np.arange(len(body_global))
        # a kinematic body referenced from several ranks is copied to each; the rank of the lowest island that references it... simply: the lowest rank using it owns it
        owned = np.ones(len(body_global), dtype=bool)
        batches = []
        for batch, brow in zip(lanes, rows):
            tbs = []
            for (tb, r), mine in zip(batch, brow):
                if len(mine) == 0:
                    continue
                nb, pf, imf, _ = TYPE_TABLE[tb.type_id]
                rr = r[mine]
                remapped = (local_of[rr & REF_MASK] | (rr & KINEMATIC_BIT)).astype(np.int32)
                tbs.append(TypeBatchData(tb.type_id, len(mine), to_aosoa(remapped, w, fill=-1), to_aosoa(tb.prestep_lanes(w)[mine], w), to_aosoa(tb.accumulated_lanes(w)[mine], w)))
            batches.append(tbs)
        handles = scene.index_to_handle[body_global].astype(np.int32)
        handle_to_index = np.full(int(scene.handle_to_index.size), -1, dtype=np.int32)
        handle_to_index[handles] = np.arange(len(body_global), dtype=np.int32)
        kin_handles = np.asarray([h for h in scene.constrained_kinematic_handles if handle_to_index[h] >= 0 and used[scene.handle_to_index[h]] and
                                  _kinematic_is_referenced(rows, lanes, int(scene.handle_to_index[h]))], dtype=np.int32)
        sub = Scene(np.ascontiguousarray(scene.bodies[body_global]), handles, handle_to_index, batches, kin_handles, w)
        shares.append(IslandShare(sub, body_global, owned, rows))
    for rank, sh in zip((range(world_size) if only_rank < 0 else [only_rank]), shares):
        sh.owned = ~(kinematic[sh.body_global] & constrained[sh.body_global] & (lowest_user[sh.body_global] != rank))
    return shares


def _kinematic_is_referenced(rows, lanes, body_index: int) -> bool:
    for batch, brow in zip(lanes, rows):
        for (tb, r), mine in zip(batch, brow):
            if len(mine) and np.any((r[mine] & REF_MASK) == body_index):
                return True
    return False


def merge_island_shares(scene, shares: Sequence[IslandShare]) -> None:
    """Write the shares' results back into the whole scene's buffers (bodies, accumulated impulses, prestep depths)."""
    from .scene import TYPE_TABLE, to_aosoa
    w = scene.bundle_width
    for sh in shares:
        scene.bodies[sh.body_global[sh.owned]] = sh.scene.bodies[sh.owned]
    for bi, batch in enumerate(scene.batches):
        for ti, tb in enumerate(batch):
            acc, pre = tb.accumulated_lanes(w), tb.prestep_lanes(w)
            for sh in shares:
                mine = sh.constraint_rows[bi][ti]
                if len(mine) == 0:
                    continue
                sub_tb = _find_type_batch(sh, bi, ti, scene)
                acc[mine] = sub_tb.accumulated_lanes(w)
                pre[mine] = sub_tb.prestep_lanes(w)
            tb.accumulated[:] = to_aosoa(acc, w)
            tb.prestep[:] = to_aosoa(pre, w)


def _find_type_batch(share: IslandShare, batch_index: int, type_batch_index: int, scene):
    """The share's type batch that came from (batch_index, type_batch_index) of the whole scene: empty parts were skipped when the share was built."""
    k = sum(1 for t in range(type_batch_index) if len(share.constraint_rows[batch_index][t]) > 0)
    return share.scene.batches[batch_index][k]
