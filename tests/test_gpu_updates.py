"""-m gpu: device-resident ranged updates through the C ABI.

Between frames the reference's narrow phase rewrites contact prestep data / accumulated impulses in place and user code rewrites individual
bodies (NarrowPhaseConstraintUpdate.cs:147-207). Patching the device copy by range must be indistinguishable, bit for bit, from uploading the
modified scene from scratch — on both schedules (the island schedule stores the constraints permuted)."""
import numpy as np
import pytest

import parity_util as pu
import small_scenes
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks, SolveDescription

pytestmark = pytest.mark.gpu

W = 8


def _mutate_between_frames(scene, rng):
    """What a frame boundary does to persisting state: new contact depths / offsets for some bundle ranges, redistributed impulses, a few kicked bodies.
    Returns the list of edits as (kind, batch, type_id, first_bundle, bundles) / ('bodies', first, rows)."""
    edits = []
    for bi, batch in enumerate(scene.batches):
        for tb in batch:
            bundles = (tb.count + W - 1) // W
            if bundles == 0 or not TYPE_TABLE[tb.type_id][3].startswith("Contact"):
                continue
            first = int(rng.integers(0, bundles))
            n = int(rng.integers(1, bundles - first + 1))
            pf, imf = tb.prestep_floats, tb.impulse_floats
            pre = tb.prestep.reshape(bundles, pf * W)
            acc = tb.accumulated.reshape(bundles, imf * W)
            pre[first:first + n] += rng.uniform(-0.01, 0.01, size=(n, pf * W)).astype(np.float32) * (pre[first:first + n] != 0)
            acc[first:first + n] *= np.float32(0.5)
            edits.append(("prestep", bi, tb.type_id, first, pre[first:first + n].copy()))
            edits.append(("impulses", bi, tb.type_id, first, acc[first:first + n].copy()))
    for _ in range(3):
        first = int(rng.integers(0, scene.body_count - 4))
        rows = scene.bodies[first:first + 4]
        dynamic = rows[:, 22] != 0
        rows[dynamic, 8:11] += rng.uniform(-0.2, 0.2, size=(int(dynamic.sum()), 3)).astype(np.float32)
        edits.append(("bodies", first, rows.copy()))
    return edits


@pytest.mark.parametrize("use_clusters", [True, False])
def test_ranged_updates_equal_a_fresh_upload(hip_solver_factory, use_clusters):
    rng = np.random.default_rng(17)
    scene = small_scenes.island_scene(5, islands=60, bodies_per_island=12, constraints_per_island=40, type_ids=[0, 3, 4, 5, 6, 7, 10, 17, 22, 30])
    sd, cb = SolveDescription(2, 4), PoseIntegratorCallbacks()
    patched = hip_solver_factory(use_clusters=use_clusters)
    patched.upload(scene, sd.fallback_batch_threshold)
    patched.solve(1 / 60, sd, cb)
    state = scene.copy()
    patched.download(state)                       # the host's view after frame 1
    edits = _mutate_between_frames(state, rng)    # ... edited in place, as the narrow phase / user code would
    for e in edits:
        if e[0] == "bodies":
            patched.update_bodies(e[1], e[2])
        elif e[0] == "prestep":
            patched.update_prestep(e[1], e[2], e[3], e[4])
        else:
            patched.update_accumulated_impulses(e[1], e[2], e[3], e[4])
    # ranged read-backs return exactly what was written (and only that range)
    for e in edits:
        if e[0] == "prestep":
            assert np.array_equal(patched.get_prestep_range(e[1], e[2], e[3], e[4].shape[0]).view(np.int32), e[4].reshape(-1).view(np.int32))
        elif e[0] == "impulses":
            assert np.array_equal(patched.get_accumulated_impulses_range(e[1], e[2], e[3], e[4].shape[0]).view(np.int32), e[4].reshape(-1).view(np.int32))
    patched.solve(1 / 60, sd, cb)
    got = state.copy()
    patched.download(got)

    fresh = hip_solver_factory(use_clusters=use_clusters)
    want = pu.run_hip(fresh, state, 1 / 60, sd, cb)          # full upload of the edited host state, one frame
    oracle = pu.run_oracle(state, 1 / 60, sd, cb, threads=4)
    for ref in (want, oracle):
        m = pu.compare_scenes(ref, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
    assert np.array_equal(patched.get_bodies_range(3, 5).view(np.int32), patched.get_bodies(scene.body_count)[3:8].view(np.int32))


def test_ranged_update_argument_checks(hip_solver_factory):
    from bepuphysics2_amd import native
    solver = hip_solver_factory()
    scene = small_scenes.random_graph_scene(3, 40, 90, [7, 22])
    with pytest.raises(native.BepuHipError):      # nothing uploaded yet
        solver.update_prestep(0, 7, 0, np.zeros(26 * W, np.float32))
    solver.upload(scene, 64)
    tb = next(t for t in scene.batches[0] if t.type_id == 7)
    bundles = (tb.count + W - 1) // W
    with pytest.raises(ValueError):               # range runs past the type batch (INVALID_ARGUMENT, the reference's ArgumentException)
        solver.update_prestep(0, 7, bundles, np.zeros(26 * W, np.float32))
    with pytest.raises(ValueError):               # no such type batch
        solver.update_prestep(0, 46, 0, np.zeros(14 * W, np.float32))
    with pytest.raises(ValueError):
        solver.update_bodies(scene.body_count - 1, np.zeros((2, 32), np.float32))
    with pytest.raises(ValueError):               # not a whole number of bundles (caught before the ABI)
        solver.update_prestep(0, 7, 0, np.zeros(26 * W + 1, np.float32))


def test_registered_memory_async_frame_and_pose_velocity_read_back(hip_solver_factory):
    """The resident frame a host runs: buffers registered once, contact prestep refreshed asynchronously, solve_async, poses and velocities read back
    asynchronously into the host's BodyDynamics array, ONE sync. Equal to the synchronous calls and to the oracle; the inertia half of the host array is left alone."""
    import oracle_ffi
    from bepuphysics2_amd.scene import TYPE_TABLE
    scene = small_scenes.random_graph_scene(57, 900, 2600, [4, 5, 6, 7, 22, 25, 30])
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory()
    host_bodies = scene.bodies.copy()
    solver.register_host_memory(host_bodies)
    contact_tbs = [(bi, tb) for bi, b in enumerate(scene.batches) for tb in b if TYPE_TABLE[tb.type_id][3].startswith("Contact")]
    for _, tb in contact_tbs:
        solver.register_host_memory(tb.prestep)
    solver.upload(scene)
    ref = scene.copy()
    rng = np.random.default_rng(3)
    for frame in range(3):
        for (bi, tb), (_, rtb) in zip(contact_tbs, [(bi, tb) for bi, b in enumerate(ref.batches) for tb in b if TYPE_TABLE[tb.type_id][3].startswith("Contact")]):
            tb.prestep[...] = rtb.prestep  # what the device holds (depths advanced by the last solve) ...
            w, pf = scene.bundle_width, tb.prestep_floats
            lane = np.arange(tb.count)
            tb.prestep[(lane // w) * pf * w + 3 * w + lane % w] += rng.uniform(-0.002, 0.002, tb.count).astype(np.float32)  # ... with the first contact's depth rewritten by the "narrow phase"
            rtb.prestep[...] = tb.prestep
            solver.update_prestep(bi, tb.type_id, 0, tb.prestep, asynchronous=True)
        solver.solve(1 / 60, sd, cb, asynchronous=True)
        sentinel = host_bodies[:, 16:].copy()
        solver.get_poses_and_velocities(host_bodies, asynchronous=True)
        solver.sync()
        oracle_ffi.solve(ref, 1 / 60, sd, cb)
        full = solver.get_bodies(scene.body_count)
        assert np.array_equal(host_bodies[:, :16].view(np.int32), full[:, :16].view(np.int32))
        assert np.array_equal(host_bodies[:, 16:].view(np.int32), sentinel.view(np.int32))  # the inertia half is the host's
        cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
        assert np.array_equal(ref.bodies[:, cols].view(np.int32), host_bodies[:, cols].view(np.int32)), frame
    solver.unregister_host_memory(host_bodies)
