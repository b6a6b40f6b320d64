"""-m gpu: BASELINE.json configs[4] on the device — one connected ragdoll lattice split into shares, solved through bepuhip_solve_exchanged with a
boundary exchange after every pass. Two processes share the box's single GPU and talk over gloo (RCCL needs one GPU per rank); the exchange code
path is the one a multi-GPU node runs with backend nccl."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
VEL = [8, 9, 10, 12, 13, 14]


def _lattice_scene(ragdolls, seed=5):
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 1, seed)
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    return scene, sd


def test_single_share_through_solve_exchanged_is_bit_exact(hip_solver_factory):
    """world = 1: no boundary, the exchange hook is called after every pass and changes nothing; result == oracle == plain bepuhip_solve."""
    import parity_util as pu
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    scene, sd = _lattice_scene(40)
    cb = PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=2, threads=4)
    share = lattice.make_share(scene, lattice.owner_by_groups(scene, 1, 16), 0, 1)
    ex = lattice.BoundaryExchange(share)
    lattice.solve_share_hip(hip_solver_factory(use_clusters=False), share, 1 / 60, sd, cb, ex, frames=2)
    assert ex.calls == 2 * int((1 + sd.iterations()).sum())
    merged = lattice.merge_owned(scene, [share])
    m = pu.compare_scenes(ref, merged)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
    with pytest.raises(Exception):  # the island schedule owns whole islands; a split scene must run the launch-per-batch schedule
        lattice.solve_share_hip(hip_solver_factory(use_clusters=True), share, 1 / 60, sd, cb, ex)


def _worker(rank, world, port, outdir, ragdolls, device_buffers):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    scene, sd = _lattice_scene(ragdolls)
    share = lattice.make_share(scene, lattice.owner_by_groups(scene, world, 16), rank, world)
    solver = HipSolver(device=0, use_clusters=False)
    ex = lattice.BoundaryExchange(share, dist, device="cuda:0" if device_buffers else None)
    lattice.solve_share_hip(solver, share, 1 / 60, sd, PoseIntegratorCallbacks(), ex, frames=2, device_buffers=device_buffers)
    solver.close()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), bodies=share.scene.bodies, l2g=share.local_to_global, owned=share.owned,
             boundary_local=share.boundary_local, boundary_slot=share.boundary_slot)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_buffers", [False, True])
def test_two_ranks_split_lattice_on_the_gpu(tmp_path, device_buffers):
    import oracle_ffi
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    ragdolls = 200
    port = 29700 + (os.getpid() % 200) + (50 if device_buffers else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path), ragdolls, device_buffers), nprocs=2, join=True)
    scene, sd = _lattice_scene(ragdolls)
    ref = scene.copy()
    for _ in range(2):
        oracle_ffi.solve(ref, 1 / 60, sd, PoseIntegratorCallbacks(), threads=4)
    got = scene.bodies.copy()
    r = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(2)]
    for d in r:
        got[d["l2g"][d["owned"]], :16] = d["bodies"][d["owned"], :16]
    rows = {}
    for d in r:
        for loc, slot in zip(d["boundary_local"], d["boundary_slot"]):
            rows.setdefault(int(slot), []).append(d["bodies"][loc])
    assert rows and all(len(v) == 2 for v in rows.values())
    for v in rows.values():  # every copy of a boundary body ends the step bit-identical (pose and velocity)
        assert np.array_equal(v[0][:15].view(np.int32), v[1][:15].view(np.int32))
    scale = float(np.abs(ref.bodies[:-1][:, VEL]).max())
    per_body = np.abs(ref.bodies[:-1][:, VEL] - got[:-1][:, VEL]).max(axis=1) / scale
    assert np.isfinite(got).all()
    assert per_body.max() < 0.05, per_body.max()       # block-Jacobi with mass splitting at the cut: a few percent on the boundary bodies
    assert np.median(per_body) < 1e-3, np.median(per_body)  # and close agreement away from it
