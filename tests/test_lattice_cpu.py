"""CPU tests of the split-scene path (BASELINE.json configs[4]): shares / ghosts / boundary exchange, with the oracle standing in for the GPU
and gloo standing in for RCCL. The GPU version of the same exchange is tests/test_gpu_lattice.py."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
VEL = [8, 9, 10, 12, 13, 14]


def _lattice_scene(ragdolls=24, seed=5):
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 1, seed)  # with contacts, chained into one lattice
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    return scene, sd


def test_shares_cover_the_scene_and_keep_batches_conflict_free():
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.scene import BODY_REFERENCE_MASK, KINEMATIC_MASK
    scene, _ = _lattice_scene()
    world = 3
    owner = lattice.owner_by_groups(scene, world, 16)
    assert (owner[:-1] >= 0).all() and owner[-1] == -1  # 24 ragdolls x 16 dynamic bodies, then the kinematic tube
    shares = [lattice.make_share(scene, owner, r, world) for r in range(world)]
    boundary = lattice.boundary_bodies(scene, owner)
    assert boundary.size > 0  # the lattice links cross the cuts
    for bi, batch in enumerate(scene.batches):
        for ti, tb in enumerate(batch):
            taken = np.concatenate([sh.constraint_source[bi][ti] for sh in shares])
            assert sorted(taken.tolist()) == list(range(tb.count))  # every constraint on exactly one rank
    for sh in shares:
        assert sh.boundary_total == boundary.size
        assert np.array_equal(sh.local_to_global[sh.boundary_local], boundary[sh.boundary_slot])
        for batch in sh.scene.batches:
            seen = set()
            for tb in batch:
                refs = tb.refs_lanes()
                dyn = refs[(refs & KINEMATIC_MASK) == 0] & BODY_REFERENCE_MASK
                assert len(set(dyn.tolist()) & seen) == 0 and len(set(dyn.tolist())) == dyn.size  # Solver.cs:1046-1051 invariant survives the split
                seen |= set(dyn.tolist())
                assert (refs & BODY_REFERENCE_MASK).max(initial=0) < sh.scene.body_count
    # every boundary body is held by its owner and at least one other rank
    holders = np.zeros(boundary.size, dtype=int)
    for sh in shares:
        holders[sh.boundary_slot] += 1
    assert (holders >= 2).all()


def test_single_rank_share_reproduces_the_plain_oracle_bit_for_bit():
    import oracle_ffi
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    from oracle_share import OracleShare
    scene, sd = _lattice_scene(12)
    cb = PoseIntegratorCallbacks()
    ref = scene.copy()
    oracle_ffi.solve(ref, 1 / 60, sd, cb)
    share = lattice.make_share(scene, lattice.owner_by_groups(scene, 1, 16), 0, 1)
    assert share.boundary_total == 0 and share.owned.sum() == scene.body_count - 1
    ex = lattice.BoundaryExchange(share)
    OracleShare(share, 1 / 60, sd, cb, ex).solve(oracle_ffi.solve)
    assert ex.calls == int((1 + sd.iterations()).sum())  # one exchange point per pass
    merged = lattice.merge_owned(scene, [share])
    assert np.array_equal(merged.bodies[:-1, :16].view(np.int32), ref.bodies[:-1, :16].view(np.int32))  # pose + velocity


def _worker(rank, world, port, outdir, ragdolls):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    from oracle_share import OracleShare
    scene, sd = _lattice_scene(ragdolls)
    share = lattice.make_share(scene, lattice.owner_by_groups(scene, world, 16), rank, world)
    ex = lattice.BoundaryExchange(share, dist)
    OracleShare(share, 1 / 60, sd, PoseIntegratorCallbacks(), ex).solve(oracle_ffi.solve, frames=2)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), bodies=share.scene.bodies, l2g=share.local_to_global, owned=share.owned,
             boundary_local=share.boundary_local, boundary_slot=share.boundary_slot)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_split_lattice(tmp_path):
    """Two ranks, one connected lattice: all copies of a boundary body end bit-identical, and the split solve stays within a few percent of the
    unsplit one (block-Jacobi across the cut is a different iteration, not a different answer)."""
    import oracle_ffi
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    ragdolls = 24
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, str(tmp_path), ragdolls), nprocs=2, join=True)
    scene, sd = _lattice_scene(ragdolls)
    ref = scene.copy()
    for _ in range(2):
        oracle_ffi.solve(ref, 1 / 60, sd, PoseIntegratorCallbacks())
    got = scene.bodies.copy()
    r = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(2)]
    for d in r:
        got[d["l2g"][d["owned"]]] = d["bodies"][d["owned"]]
    # copies of the same boundary body
    rows = {}
    for d in r:
        for loc, slot in zip(d["boundary_local"], d["boundary_slot"]):
            rows.setdefault(int(slot), []).append(d["bodies"][loc])
    assert rows and all(len(v) == 2 for v in rows.values())
    for v in rows.values():
        assert np.array_equal(v[0][:15].view(np.int32), v[1][:15].view(np.int32))
    dyn = slice(0, scene.body_count - 1)
    scale = float(np.abs(ref.bodies[dyn][:, VEL]).max())
    err = float(np.abs(ref.bodies[dyn][:, VEL] - got[dyn][:, VEL]).max()) / scale
    assert np.isfinite(got).all()
    assert err < 0.05, err
    interior = np.ones(scene.body_count, dtype=bool)
    interior[-1] = False
    far = np.abs(ref.bodies[:, VEL] - got[:, VEL]).max(axis=1) / scale
    assert np.median(far[interior]) < 1e-3  # away from the cut the two solves agree closely


def _exact_worker(rank, world, port, outdir):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bepuphysics2_amd import lattice
    scene, _ = _lattice_scene(24)
    share = lattice.make_share(scene, lattice.owner_by_groups(scene, world, 16), rank, world, mass_split=False)
    ex = lattice.BoundaryExchange(share, dist)
    rng = np.random.default_rng(7)  # the same stream on both ranks: a "solved" value for every dense row, and which rank touched it in this batch
    truth = rng.integers(0, 2**32, size=(share.boundary_total, 6), dtype=np.uint64).astype(np.uint32)
    toucher = rng.integers(0, world + 1, size=share.boundary_total)  # == world: nobody
    snapshot = rng.integers(0, 2**32, size=(share.boundary_total, 6), dtype=np.uint64).astype(np.uint32)
    mine = toucher[share.boundary_slot] == rank
    local_now = snapshot[share.boundary_slot].copy()
    local_now[mine] = truth[share.boundary_slot][mine]
    total = ex.reduce_exact(local_now ^ snapshot[share.boundary_slot])
    np.savez(os.path.join(outdir, f"exact{rank}.npz"), new=snapshot[share.boundary_slot] ^ total, slot=share.boundary_slot, truth=truth, toucher=toucher, snapshot=snapshot)
    dist.barrier()
    dist.destroy_process_group()


def test_exact_mode_exchange_moves_the_touchers_bit_pattern_over_gloo(tmp_path):
    """BEPUHIP_EXCHANGE_PER_BATCH_EXACT's transport: XOR patterns, integer sum. Whatever one rank wrote arrives bit for bit on every holder, rows nobody
    touched keep their pattern (NaN payloads and negative zeros included: these are raw words)."""
    mp.spawn(_exact_worker, args=(2, 29300 + (os.getpid() % 300), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        d = np.load(os.path.join(str(tmp_path), f"exact{rank}.npz"))
        want = np.where((d["toucher"] < 2)[:, None], d["truth"], d["snapshot"])[d["slot"]]
        assert d["slot"].size and np.array_equal(d["new"], want)


def test_full_mass_shares_for_the_exact_mode():
    from bepuphysics2_amd import lattice
    scene, _ = _lattice_scene(12)
    owner = lattice.owner_by_groups(scene, 2, 16)
    split, full = lattice.make_share(scene, owner, 0, 2), lattice.make_share(scene, owner, 0, 2, mass_split=False)
    g = full.local_to_global
    assert np.array_equal(full.scene.bodies[:, 16:31], scene.bodies[g, 16:31])
    b = split.boundary_local
    assert b.size and np.allclose(split.scene.bodies[b, 22], 2 * scene.bodies[g[b], 22])


def test_oracle_shares_in_threads_equal_the_gloo_processes(tmp_path):
    """The in-process CPU lattice (oracle_share.solve_oracle_shares_in_process: one thread per rank, ThreadExchange) is the oracle tests/test_gpu_lattice.py holds the device's
    block-Jacobi shares against; here it is checked against the two-process gloo run of the same shares: same sums, same bits."""
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    from oracle_share import solve_oracle_shares_in_process
    ragdolls = 24
    port = 29100 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, str(tmp_path), ragdolls), nprocs=2, join=True)
    scene, sd = _lattice_scene(ragdolls)
    owner = lattice.owner_by_groups(scene, 2, 16)
    shares = [lattice.make_share(scene, owner, r, 2) for r in range(2)]
    ex = solve_oracle_shares_in_process(shares, 1 / 60, sd, PoseIntegratorCallbacks(), frames=2)
    assert ex.calls == 2 * int((1 + sd.iterations()).sum())
    for rank, sh in enumerate(shares):
        d = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(sh.scene.bodies[:, :15].view(np.int32), d["bodies"][:, :15].view(np.int32)), rank
