"""world_size-2 gloo test of the N>1 path: islands are sharded whole across ranks, each rank solves only its own islands
(here with the CPU oracle standing in for the GPU), and the union equals the single-process result bit for bit;
throughput aggregation = sum(units) / max(elapsed)."""
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
    # 6 islands (ragdolls) of equal size -> 3 per rank
    parts = sharding.partition_islands([58] * 6, world)
    sim = HostSimulation.scene("ragdoll_tube", len(parts[rank]), 0, 0, sharding.rank_seed(5, rank))
    scene, sd = sim.export(), sim.solve_description()
    oracle_ffi.solve(scene, 1 / 60, sd, PoseIntegratorCallbacks())
    units = scene.constraint_count * int((1 + sd.iterations()).sum())
    total = sharding.aggregate_throughput(dist, units, 0.5 + rank)  # fake elapsed: rank 1 is slower
    dist.barrier()
    if rank == 0:
        np.save(out, np.asarray([total, units], dtype=np.float64))
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
    total, units = np.load(out)
    assert abs(total - 2 * units / 1.5) < 1e-6  # sum of units over ranks / max elapsed (1.5 s on rank 1)
