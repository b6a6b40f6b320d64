"""One connected scene split across GPUs (BASELINE.json configs[4]): shares, ghost bodies and the boundary exchange.

Bodies are partitioned by owner rank. A constraint belongs to the owner of its first dynamic body; the other body of a cut constraint
becomes a *ghost* on that rank. Every rank solves its share with the ordinary batches (a subset of a valid colouring is a valid colouring,
so the global batch indices are kept), and after every pass the ranks sum what their constraints did to the bodies that exist on more
than one rank and reset every copy to ``snapshot + sum / holders``: Gauss-Seidel inside a share, block-Jacobi across the cut. All copies
of a boundary body therefore stay bit-identical to each other; the result is NOT bit-identical to the unsplit solve.

Plain Jacobi on a light body (a ragdoll's hand) held by two ranks overshoots: both ranks correct the same velocity error in full and the
sum applies it twice; the copies diverge within two frames. Mass splitting (Tonge, Benevolenski, Voroshilov: "Mass splitting for
jitter-free parallel rigid body simulation", SIGGRAPH 2012) removes that: every copy of a body with k holders solves against 1/k of
its mass (inverse mass and inverse inertia x k) and the ranks' deltas are averaged, so the summed impulse acts on the full mass while no
rank can correct more than its share.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .scene import BODY_REFERENCE_MASK, KINEMATIC_MASK, Scene, TypeBatchData, to_aosoa


@dataclass
class Share:
    """What one rank uploads and how it maps back to the unsplit scene."""
    rank: int
    world: int
    scene: Scene                      # local bodies (owned first, then ghosts and replicated kinematics) + the constraints assigned to this rank
    local_to_global: np.ndarray       # int64 [local body count]
    owned: np.ndarray                 # bool  [local body count]
    boundary_local: np.ndarray        # int32 local indices of the boundary bodies this rank holds
    boundary_slot: np.ndarray         # int64 their rows in the dense exchange buffer (same row for the same body on every rank)
    boundary_total: int               # rows of the dense exchange buffer
    boundary_holders: np.ndarray      # float32 [boundary_total]: number of ranks holding each boundary body (mass-splitting factor)
    constraint_source: List[List[np.ndarray]]  # per batch, per type batch: indices into the unsplit type batch


def owner_by_position(scene: Scene, world: int, axis: int = 0) -> np.ndarray:
    """Slab partition along one axis with (almost) equal body counts per rank; kinematic bodies get owner -1 (replicated)."""
    pos = scene.bodies[:, 4 + axis]
    dynamic = np.any(scene.bodies[:, 16:23] != 0, axis=1)
    order = np.argsort(pos[dynamic], kind="stable")
    owner = np.full(scene.body_count, -1, dtype=np.int64)
    dyn_idx = np.nonzero(dynamic)[0][order]
    owner[dyn_idx] = (np.arange(dyn_idx.size) * world) // max(dyn_idx.size, 1)
    return owner


def owner_by_groups(scene: Scene, world: int, group_size: int) -> np.ndarray:
    """Consecutive groups of ``group_size`` bodies (e.g. 16 = one ragdoll) in creation order, whole groups per rank."""
    dynamic = np.any(scene.bodies[:, 16:23] != 0, axis=1)
    groups = np.arange(scene.body_count) // group_size
    n_groups = int(groups.max()) + 1 if scene.body_count else 0
    owner = (groups * world) // max(n_groups, 1)
    return np.where(dynamic, owner, -1).astype(np.int64)


def _constraint_owner(refs: np.ndarray, owner: np.ndarray) -> np.ndarray:
    """Rank of every constraint: owner of its first dynamic body."""
    idx = refs & BODY_REFERENCE_MASK
    kin = (refs & KINEMATIC_MASK) != 0
    own = np.where(kin, -1, owner[idx])
    out = own[:, 0].copy()
    for k in range(1, refs.shape[1]):
        out = np.where(out < 0, own[:, k], out)
    if (out < 0).any():
        raise ValueError("a constraint without a dynamic body cannot be assigned to a rank")
    return out


def boundary_holders(scene: Scene, owner: np.ndarray, world: int) -> np.ndarray:
    """[body, rank] bool: which ranks hold a copy of each dynamic body (its owner, and every rank one of whose constraints references it)."""
    held = np.zeros((scene.body_count, world), dtype=bool)
    dyn_bodies = np.nonzero(owner >= 0)[0]
    held[dyn_bodies, owner[dyn_bodies]] = True
    for batch in scene.batches:
        for tb in batch:
            refs = tb.refs_lanes(scene.bundle_width).astype(np.int64)
            c_owner = _constraint_owner(refs, owner)
            idx = refs & BODY_REFERENCE_MASK
            dyn = (refs & KINEMATIC_MASK) == 0
            for k in range(refs.shape[1]):
                held[idx[dyn[:, k], k], c_owner[dyn[:, k]]] = True
    return held


def boundary_bodies(scene: Scene, owner: np.ndarray) -> np.ndarray:
    """Global indices (ascending) of the dynamic bodies referenced by a constraint assigned to a rank other than their owner."""
    marks = np.zeros(scene.body_count, dtype=bool)
    for batch in scene.batches:
        for tb in batch:
            refs = tb.refs_lanes(scene.bundle_width).astype(np.int64)
            c_owner = _constraint_owner(refs, owner)
            idx = refs & BODY_REFERENCE_MASK
            dyn = (refs & KINEMATIC_MASK) == 0
            foreign = dyn & (owner[idx] != c_owner[:, None])
            marks[idx[foreign]] = True
    return np.nonzero(marks)[0]


def make_share(scene: Scene, owner: np.ndarray, rank: int, world: int, mass_split: bool = True) -> Share:
    """``mass_split=False``: copies keep their full masses — the shares of the per-batch exact exchange (bepuhip.h BEPUHIP_EXCHANGE_PER_BATCH_EXACT)."""
    w = scene.bundle_width
    boundary = boundary_bodies(scene, owner)
    needed = owner == rank
    picks: List[List[np.ndarray]] = []
    for batch in scene.batches:
        row = []
        for tb in batch:
            refs = tb.refs_lanes(w).astype(np.int64)
            mine = np.nonzero(_constraint_owner(refs, owner) == rank)[0]
            row.append(mine)
            needed[(refs[mine] & BODY_REFERENCE_MASK).reshape(-1)] = True
        picks.append(row)
    owned_idx = np.nonzero(owner == rank)[0]
    other_idx = np.nonzero(needed & (owner != rank))[0]  # ghosts and replicated kinematics
    local_to_global = np.concatenate([owned_idx, other_idx]).astype(np.int64)
    global_to_local = np.full(scene.body_count, -1, dtype=np.int64)
    global_to_local[local_to_global] = np.arange(local_to_global.size)
    batches: List[List[TypeBatchData]] = []
    for batch, row in zip(scene.batches, picks):
        out_row = []
        for tb, mine in zip(batch, row):
            refs = tb.refs_lanes(w).astype(np.int64)[mine]
            local_refs = (global_to_local[refs & BODY_REFERENCE_MASK] | (refs & KINEMATIC_MASK)).astype(np.int32)
            out_row.append(TypeBatchData(tb.type_id, int(mine.size), to_aosoa(local_refs.reshape(-1, tb.bodies), w, fill=-1),
                                         to_aosoa(tb.prestep_lanes(w)[mine].reshape(-1, tb.prestep_floats), w),
                                         to_aosoa(tb.accumulated_lanes(w)[mine].reshape(-1, tb.impulse_floats), w)))
        batches.append(out_row)
    n_local = local_to_global.size
    handles = np.arange(n_local, dtype=np.int32)  # local handle == local index
    kin_global = scene.handle_to_index[scene.constrained_kinematic_handles]
    kin_local = global_to_local[kin_global]
    local_scene = Scene(scene.bodies[local_to_global].copy(), handles.copy(), handles.copy(), batches,
                        kin_local[kin_local >= 0].astype(np.int32), w)
    held = global_to_local[boundary] >= 0
    holders = boundary_holders(scene, owner, world)[boundary].sum(axis=1).astype(np.float32)
    # mass splitting: this rank's copy of a body with k holders carries 1/k of its mass
    b_local = global_to_local[boundary][held]
    if mass_split:
        local_scene.bodies[b_local, 16:23] *= holders[held][:, None]
        local_scene.bodies[b_local, 24:31] *= holders[held][:, None]
    return Share(rank, world, local_scene, local_to_global, np.arange(n_local) < owned_idx.size,
                 b_local.astype(np.int32), np.nonzero(held)[0].astype(np.int64), int(boundary.size), holders, picks)


def merge_owned(unsplit: Scene, shares: List[Share]) -> Scene:
    """Assemble the unsplit scene's buffers from every rank's owned bodies and its constraints' impulses / prestep."""
    out = unsplit.copy()
    w = unsplit.bundle_width
    for sh in shares:
        g = sh.local_to_global[sh.owned]
        out.bodies[g, :16] = sh.scene.bodies[sh.owned, :16]  # pose and velocity; the inertias stay the unsplit scene's (shares carry split masses)
        kinematic = ~np.any(sh.scene.bodies[:, 16:23] != 0, axis=1) & ~sh.owned  # replicated kinematic bodies advance identically on every rank
        out.bodies[sh.local_to_global[kinematic], :16] = sh.scene.bodies[kinematic, :16]
    for bi, batch in enumerate(out.batches):
        for ti, tb in enumerate(batch):
            acc, pre = tb.accumulated_lanes(w), tb.prestep_lanes(w)
            for sh in shares:
                mine = sh.constraint_source[bi][ti]
                if mine.size:
                    acc[mine] = sh.scene.batches[bi][ti].accumulated_lanes(w)
                    pre[mine] = sh.scene.batches[bi][ti].prestep_lanes(w)
            tb.accumulated[...] = to_aosoa(acc.reshape(-1, tb.impulse_floats), w)
            tb.prestep[...] = to_aosoa(pre.reshape(-1, tb.prestep_floats), w)
    return out


class BoundaryExchange:
    """The per-pass exchange: ``deltas()`` -> dense buffer -> all-reduce(sum) -> ``apply(sums)``. ``dist`` is torch.distributed (RCCL on
    GPUs via device tensors, gloo with host staging in the CPU tests) or None for a single rank (the exchange then only re-bases)."""

    def __init__(self, share: Share, dist=None, device: Optional[str] = None):
        self.share, self.dist, self.device = share, dist, device
        self.calls = 0

    def reduce(self, local_deltas: np.ndarray) -> np.ndarray:
        """[held boundary bodies, 6] -> the summed rows for the same bodies."""
        self.calls += 1
        if self.share.boundary_total == 0:
            return local_deltas
        dense = np.zeros((self.share.boundary_total, 6), dtype=np.float32)
        dense[self.share.boundary_slot] = local_deltas
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            import torch
            t = torch.from_numpy(dense)
            if self.device is not None:
                t = t.to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            dense = t.cpu().numpy()
        return self.average(dense)[self.share.boundary_slot]

    def reduce_exact(self, local_patterns: np.ndarray) -> np.ndarray:
        """Per-batch exact mode: [held boundary bodies, 6] uint32 XOR patterns (at most one rank's row is non-zero for any body) -> the summed patterns."""
        self.calls += 1
        if self.share.boundary_total == 0:
            return local_patterns
        dense = np.zeros((self.share.boundary_total, 6), dtype=np.int32)
        dense[self.share.boundary_slot] = np.ascontiguousarray(local_patterns).view(np.int32)
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            import torch
            t = torch.from_numpy(dense)  # int32: wrap-around addition, exact with a single non-zero contribution
            if self.device is not None:
                t = t.to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            dense = t.cpu().numpy()
        return dense[self.share.boundary_slot].view(np.uint32)

    def average(self, dense_sums: np.ndarray) -> np.ndarray:
        """Mass splitting: the copies solved against 1/k of the mass, so the mean of their velocity changes is the full-mass response."""
        return (dense_sums / self.share.boundary_holders[:, None]).astype(np.float32)


def solve_share_hip(solver, share: Share, dt, solve_description, callbacks, exchange: BoundaryExchange, frames: int = 1, upload: bool = True,
                    device_buffers: bool = False, exact: bool = False):
    """Run ``frames`` steps of this rank's share on the GPU (HipSolver created with use_clusters=False), exchanging after every pass — or, ``exact``, after
    every batch (the share must then come from make_share(mass_split=False)).
    ``device_buffers``: keep the deltas in HBM and all-reduce device tensors (RCCL over xGMI on a multi-GPU node); otherwise stage through the host."""
    if upload:
        solver.upload(share.scene, solve_description.fallback_batch_threshold)
        solver.set_boundary_bodies(share.boundary_local)
    solver.set_exchange_mode(1 if exact else 0)
    hook = DeviceExchange(solver, share, exchange.dist, exchange.device) if device_buffers and not exact else None

    def host_hook(_substep, _pass):
        if exact:
            solver.boundary_apply(exchange.reduce_exact(solver.boundary_deltas()))
        else:
            solver.boundary_apply(exchange.reduce(solver.boundary_deltas()))

    for _ in range(frames):
        solver.solve_exchanged(dt, solve_description, callbacks, hook if hook is not None else host_hook)
    if hook is not None:
        exchange.calls += hook.calls
    solver.download(share.scene)


class DeviceExchange:
    """The exchange with every buffer resident in HBM: boundary_deltas -> scatter into the dense buffer -> all_reduce (RCCL) -> gather, divide by
    the holder count -> boundary_apply. torch is only the allocator / collective front-end here."""

    def __init__(self, solver, share: Share, dist, device):
        import torch
        self.torch, self.solver, self.share, self.dist = torch, solver, share, dist
        dev = device or "cuda:0"
        self.local = torch.zeros((max(share.boundary_local.size, 1), 6), dtype=torch.float32, device=dev)
        self.dense = torch.zeros((max(share.boundary_total, 1), 6), dtype=torch.float32, device=dev)
        self.slot = torch.from_numpy(share.boundary_slot).to(dev)
        self.holders = torch.from_numpy(share.boundary_holders).to(dev)
        self.calls = 0

    def __call__(self, _substep, _pass):
        self.calls += 1
        if self.share.boundary_total == 0:
            return
        t = self.torch
        n = self.share.boundary_local.size
        if n:
            self.solver.boundary_deltas_device(self.local.data_ptr())  # synchronises the solver's stream before returning
        self.dense.zero_()
        if n:
            self.dense[self.slot] = self.local[:n]
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            self.dist.all_reduce(self.dense, op=self.dist.ReduceOp.SUM)
        if n:
            self.local[:n] = self.dense[self.slot] / self.holders[self.slot][:, None]
            t.cuda.synchronize()
            self.solver.boundary_apply_device(self.local.data_ptr())


def solve_share_on_stream(solver, share: Share, dt, solve_description, callbacks, frames: int = 1, upload: bool = True, exact: bool = False,
                          unique_id: Optional[bytes] = None):
    """The production path of a multi-GPU node: the exchange runs on the solver's stream (bepuhip_solve_lattice), RCCL all-reduce over a communicator the library
    creates from ``unique_id`` (made by rank 0 with solver.comm_unique_id() and carried to the other ranks by the host's own transport). ``unique_id`` None: a
    single rank, no collective."""
    if upload:
        solver.upload(share.scene, solve_description.fallback_batch_threshold)
        solver.set_boundary_bodies(share.boundary_local)
        solver.set_boundary_layout(share.boundary_slot, share.boundary_total, None if exact else share.boundary_holders)
        if unique_id is not None:
            solver.comm_init(unique_id, share.rank, share.world)
    solver.set_exchange_mode(1 if exact else 0)
    for _ in range(frames):
        solver.solve_lattice(dt, solve_description, callbacks)
    solver.download(share.scene)


class ThreadExchange:
    """All ranks of a split scene inside ONE process (one thread and one HipSolver context per rank, e.g. on a single-GPU box): the ranks meet at a barrier and
    sum their rows in host memory. Same arithmetic as BoundaryExchange over a process group; used by bench.py's lattice leg and the GPU tests."""

    def __init__(self, shares: List[Share]):
        import threading
        self.shares = shares
        self.barrier = threading.Barrier(len(shares))
        total = shares[0].boundary_total
        self.dense_f = np.zeros((len(shares), max(total, 1), 6), dtype=np.float32)
        self.dense_i = np.zeros((len(shares), max(total, 1), 6), dtype=np.int32)
        self.calls = 0

    def reduce(self, rank: int, local: np.ndarray, exact: bool) -> np.ndarray:
        sh = self.shares[rank]
        if rank == 0:
            self.calls += 1
        if sh.boundary_total == 0:
            return local
        buf = self.dense_i if exact else self.dense_f
        buf[rank] = 0
        buf[rank][sh.boundary_slot] = np.ascontiguousarray(local).view(np.int32) if exact else local
        self.barrier.wait()
        if exact:
            total = buf.sum(axis=0, dtype=np.int32)  # wrap-around, exact with one non-zero contribution
            out = total[sh.boundary_slot].view(np.uint32)
        else:
            total = buf[0].copy()
            for r in range(1, len(self.shares)):  # rank order: the order a ring all-reduce is free to differ from; the tolerance of this mode covers it
                total += buf[r]
            out = (total / sh.boundary_holders[:, None]).astype(np.float32)[sh.boundary_slot]
        self.barrier.wait()  # nobody overwrites its row before everyone has read the sums
        return out


def solve_shares_in_process(make_solver, shares: List[Share], dt, solve_description, callbacks, frames: int = 1, exact: bool = False) -> ThreadExchange:
    """Solve every share of a split scene in this process, one thread per rank (ctypes releases the GIL inside the library)."""
    import threading
    ex = ThreadExchange(shares)
    errors = []

    def run(rank):
        solver = None
        try:
            solver = make_solver()
            share = shares[rank]
            solver.upload(share.scene, solve_description.fallback_batch_threshold)
            solver.set_boundary_bodies(share.boundary_local)
            solver.set_exchange_mode(1 if exact else 0)

            def hook(_substep, _pass):
                solver.boundary_apply(ex.reduce(rank, solver.boundary_deltas(), exact))

            for _ in range(frames):
                solver.solve_exchanged(dt, solve_description, callbacks, hook)
            solver.download(share.scene)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            ex.barrier.abort()
        finally:
            if solver is not None:
                solver.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(len(shares))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return ex


# ---- device groups (round 5): the exact mode on the island schedule ----
def solve_group_in_process(make_solver, scene: Scene, world: int, dt, solve_description, callbacks, frames: int = 1) -> Scene:
    """One connected scene solved by a GROUP of ``world`` contexts in this process (one thread each; e.g. all on a single-GPU box): every member uploads the whole scene,
    plans the same clusters (bepuhip_set_device_group) and runs its range of them; shared bodies cross from member to member through the event-numbered records every
    member pushes into every other member's table — no exchange point inside a step. After every frame the members' owned bodies are merged on the host (on a multi-GPU
    node: bepuhip_sync_owned_bodies, one all-reduce) and every member starts the next frame from the merged bodies. Returns the merged scene: bodies from their owners,
    every constraint's impulses and prestep data from the member whose cluster runs it. Bit-identical to the unsplit solve."""
    import threading
    barrier = threading.Barrier(world)
    records = [0] * world
    merged_bodies = scene.bodies.copy()
    results = [None] * world
    errors = []

    def run(rank):
        solver = None
        try:
            solver = make_solver()
            mine = scene.copy()
            solver.set_device_group(world, rank)
            solver.upload(mine, solve_description.fallback_batch_threshold)
            records[rank] = solver.shared_records()[0]
            barrier.wait()  # every member has planned and cleared its record table
            peers = [records[r] for r in range(world) if r != rank]
            if all(peers):
                for k, pointer in enumerate(peers):
                    solver.set_peer_records(k, pointer)
            owned = solver.owned_bodies(mine.body_count)
            barrier.wait()
            for frame in range(frames):
                if frame > 0:
                    solver.set_bodies(merged_bodies)
                    barrier.wait()  # (nobody overwrites the merged array before everybody has read it)
                solver.solve(dt, solve_description, callbacks)
                got = solver.get_bodies(mine.body_count)
                barrier.wait()
                merged_bodies[owned] = got[owned]
                barrier.wait()
            solver.download(mine)
            masks = {(bi, tb.type_id): solver.owned_constraints(bi, tb.type_id, tb.count) for bi, b in enumerate(mine.batches) for tb in b if tb.count}
            results[rank] = (mine, owned, masks, solver.schedule(), int(solver.cluster_cycles().size))
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            barrier.abort()
        finally:
            if solver is not None:
                solver.close()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    out = scene.copy()
    covered = np.zeros(scene.body_count, dtype=int)
    for mine, owned, masks, _, _ in results:
        out.bodies[owned] = mine.bodies[owned]
        covered += owned
        for bi, b in enumerate(mine.batches):
            for ti, tb in enumerate(b):
                if not tb.count:
                    continue
                lanes = masks[(bi, tb.type_id)]
                target = out.batches[bi][ti]
                w = scene.bundle_width
                for src, dst, fields in ((tb.accumulated, target.accumulated, tb.impulse_floats), (tb.prestep, target.prestep, tb.prestep_floats)):
                    s3, d3 = src.reshape(-1, fields, w), dst.reshape(-1, fields, w)
                    idx = np.nonzero(lanes)[0]
                    d3[idx // w, :, idx % w] = s3[idx // w, :, idx % w]
    assert (covered == 1).all(), "every body is owned by exactly one member of the group"
    out.group_info = [(r[3], r[4]) for r in results]
    return out
