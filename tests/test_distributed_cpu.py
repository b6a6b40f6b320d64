"""world_size-2 gloo test of the N>1 path (BASELINE.json configs[3]): ONE scene is cut by `sharding.split_scene_by_islands` into whole islands per
rank, each rank solves only its share (here with the CPU oracle standing in for the GPU), and the union of the shares' results equals the
single-process solve of the whole scene bit for bit; throughput aggregation = sum(units) / max(elapsed)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi
    from bepuphysics2_amd import sharding
    from bepuphysics2_amd.hostlib import HostSimulation
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    # ONE scene (14 ragdolls = 14 islands sharing the kinematic tube, plus ground contacts), built identically on every rank, cut by the partitioner.
    sim = HostSimulation.scene("ragdoll_tube", 14, 1, 0, 5)
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    share = sharding.split_scene_by_islands(scene, world, only_rank=rank)[0]
    cb = PoseIntegratorCallbacks()
    for _ in range(3):
        oracle_ffi.solve(share.scene, 1 / 60, sd, cb)  # the CPU oracle stands in for the GPU; no data-path collective between the frames
    units = share.scene.constraint_count * int((1 + sd.iterations()).sum()) * 3
    total = sharding.aggregate_throughput(dist, units, 0.5 + rank)  # fake elapsed: rank 1 is slower
    # bring every share's results to rank 0 (test plumbing: a real job keeps them on their GPUs)
    gathered = [None] * world
    dist.all_gather_object(gathered, (share.scene.bodies, [[(tb.accumulated, tb.prestep) for tb in b] for b in share.scene.batches]))
    if rank == 0:
        shares = sharding.split_scene_by_islands(scene, world)
        for sh, (bodies, batches) in zip(shares, gathered):
            sh.scene.bodies[:] = bodies
            for b, gb in zip(sh.scene.batches, batches):
                for tb, (acc, pre) in zip(b, gb):
                    tb.accumulated[:], tb.prestep[:] = acc, pre
        merged = scene.copy()
        sharding.merge_island_shares(merged, shares)
        ref = scene.copy()
        for _ in range(3):
            oracle_ffi.solve(ref, 1 / 60, sd, cb)
        cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
        same_bodies = np.array_equal(ref.bodies[:, cols].view(np.int32), merged.bodies[:, cols].view(np.int32))
        same_impulses = all(np.array_equal(a.accumulated_lanes().view(np.int32), b.accumulated_lanes().view(np.int32)) and
                            np.array_equal(a.prestep_lanes().view(np.int32), b.prestep_lanes().view(np.int32))
                            for ba, bb in zip(ref.batches, merged.batches) for a, b in zip(ba, bb))
        total_units = scene.constraint_count * int((1 + sd.iterations()).sum()) * 3
        np.save(out, np.asarray([total, total_units, float(same_bodies), float(same_impulses), float(sum(len(g[0]) for g in gathered))], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_deterministic():
    from bepuphysics2_amd import sharding
    parts = sharding.partition_islands([10, 3, 7, 7, 1, 5, 9, 2], 3)
    assert sorted(i for p in parts for i in p) == list(range(8))
    loads = [sum([10, 3, 7, 7, 1, 5, 9, 2][i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 3
    assert parts == sharding.partition_islands([10, 3, 7, 7, 1, 5, 9, 2], 3)


def test_connected_components_finds_ragdoll_islands():
    from bepuphysics2_amd import sharding
    from bepuphysics2_amd.hostlib import HostSimulation
    sc = HostSimulation.scene("ragdoll_tube", 5, 1, 0, 5).export()
    pairs = []
    for b in sc.batches:
        for tb in b:
            r = tb.refs_lanes().astype(np.int64)
            r = np.where((r & (1 << 30)) != 0, -1, r)  # kinematic refs never connect islands
            if r.shape[1] == 1:
                continue
            pairs.append(r)
    labels = sharding.connected_components(sc.body_count, np.concatenate(pairs))
    dyn = np.any(sc.bodies[:, 16:23] != 0, axis=1)
    assert len(set(labels[dyn].tolist())) == 5  # 5 ragdolls = 5 islands; the kinematic tube does not merge them


def test_two_rank_gloo_sharded_solve(tmp_path):
    out = str(tmp_path / "r.npy")
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    total, total_units, same_bodies, same_impulses, gathered_bodies = np.load(out)
    assert same_bodies == 1.0 and same_impulses == 1.0  # the union of the shares' results IS the single-process result, bit for bit
    assert abs(total - total_units / 1.5) < 1e-6  # sum of units over ranks / max elapsed (1.5 s on rank 1); every constraint is solved exactly once
    assert gathered_bodies >= 14 * 16 + 1
