"""The resident frame: what a host rewrites IN PLACE between frames must reach the device, and what the device changes must come back.

The C++ twin of integration/csharp/HipTimestepper.cs (host/bepu_host.cpp, HipTimestepper mode 2) runs the frame the C# shim runs: the structural diff in two phases,
a shadow of the device's joint prestep data and impulses compared against the host's buffers every frame, changed bundle ranges sent with
bepuhip_transfer_rows_async, poses / velocities / ALL accumulated impulses back asynchronously behind the solve. Every frame's result must equal the oracle's solve of the
host mirror's export of that frame, bit for bit — while the test does to the mirror what the reference's demos and its own bookkeeping do to a simulation:
  * Solver.ApplyDescription on servos and motors every frame (Demos/Demos/Tanks/Tank.cs:100,139,142; Demos/Demos/Cars/SimpleCar.cs:25; Solver.cs:1162-1185);
  * a joint island goes to sleep (its constraints leave the active set with their accumulated impulses, IslandSleeper.cs:174-260) and wakes up later (they come back WITH
    those impulses, IslandAwakener.cs:388-400);
  * Solver.Remove + Solver.Add in one frame handing the same constraint handle out again at the same index (IdPool.Take is last-in-first-out);
  * a constraint replaced by one of ANOTHER type on the same bodies in the same batch within one frame (removal and addition in different type batches)."""
import numpy as np
import pytest

import oracle_ffi
import parity_util as pu
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks

pytestmark = pytest.mark.gpu

TWIST_SERVO, ANGULAR_MOTOR, BALL_SOCKET, SWING_LIMIT = 26, 30, 22, 25
NAMES = {info[3]: t for t, info in TYPE_TABLE.items()}
REQUIRE_RESTORED_IMPULSES = True  # (a dry run of the test's host side without a device, where nothing ever accumulates, turns this off)


def constraints_of(sim, export, predicate):
    """(handle, batch, type id, index, body handles, prestep lane, impulse lane) of every live constraint `predicate(type_id, body_handles)` accepts."""
    import ctypes as C
    out = []
    for bi, tbs in enumerate(export.batches):
        for ti, tb in enumerate(tbs):
            if not tb.count:
                continue
            handles = np.ctypeslib.as_array(sim.lib.bepuhost_type_batch_handles(sim.h, bi, ti), shape=(tb.count,)).copy()
            refs, pre, acc = tb.refs_lanes(8), tb.prestep_lanes(8), tb.accumulated_lanes(8)
            for i in range(tb.count):
                bodies = [int(export.index_to_handle[int(r) & 0x3FFFFFFF]) for r in refs[i]]
                if predicate(tb.type_id, bodies):
                    out.append((int(handles[i]), bi, tb.type_id, i, bodies, pre[i].copy(), acc[i].copy()))
    return out


def check_frame(sim, sd, cb, frame, what):
    export = sim.export()
    ref = export.copy()
    oracle_ffi.solve(ref, 1 / 60, sd, cb, threads=4)
    sim.timestep(1 / 60)
    got = sim.export()
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, what, m)
    return got


@pytest.mark.parametrize("layout", ["islands", "split", "batches"])
def test_resident_frame_follows_descriptions_sleeping_islands_and_reused_handles(hip_solver_factory, monkeypatch, layout):
    from bepuphysics2_amd.hostlib import HostSimulation
    if layout == "split":
        monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "12")
        monkeypatch.setenv("BEPUHIP_FORCE_SPLIT", "64")  # the crowd's one island is cut although a workgroup could hold it
    if layout == "islands":
        monkeypatch.setenv("BEPUHIP_CLUSTER_BODIES", "256")  # three clusters of sixteen ragdolls
    if layout == "batches":
        monkeypatch.setenv("BEPUHIP_NO_CLUSTERS", "1")
    sim = HostSimulation.scene("ragdoll_tube", 48, 1, 2 if layout == "split" else 0, 7)
    sd, cb = sim.solve_description(), PoseIntegratorCallbacks()
    sim.attach_hip_timestepper(0)
    sim.timestepper_mode(2)
    sim.replan_interval(1)  # a frame whose changes the plan cannot absorb (a NEW type batch in a batch: frame 7) costs one frame on launch-per-batch rows, then a re-plan
    sim.read_back_contact_depths(True)  # the test has no narrow phase that would rewrite them: the host keeps the device's depths, as the oracle's input expects
    rng = np.random.default_rng(11)
    body_of_ragdoll = lambda r: set(range(16 * r, 16 * r + 16))  # body handles == creation order: 16 per ragdoll
    asleep = []          # (type id, body handles, prestep lane, impulse lane) of the sleeping island's constraints
    sleepers = {3, 4, 5}
    expected_schedule = {"islands": 1, "split": 2, "batches": 0}[layout]
    schedules = []
    for frame in range(14):
        export = sim.export()
        what = []
        # -- every frame: new targets for a third of the twist servos, new speeds for a third of the angular motors (Solver.ApplyDescription) --
        servos = constraints_of(sim, export, lambda t, b: t == TWIST_SERVO)
        motors = constraints_of(sim, export, lambda t, b: t == ANGULAR_MOTOR)
        for h, _, _, _, _, lane, _ in servos[frame % 3::3]:
            lane[8] = np.float32(rng.uniform(-0.6, 0.6))             # TwistServoPrestepData.TargetAngle (TwistServo.cs:77-84)
            sim.apply_description(h, lane)
        for h, _, _, _, _, lane, _ in motors[(frame + 1) % 3::3]:
            lane[0:3] = rng.uniform(-1.5, 1.5, 3).astype(np.float32)  # AngularMotorPrestepData.TargetVelocityLocalA (AngularMotor.cs:55-59)
            sim.apply_description(h, lane)
        what.append(f"{len(servos[frame % 3::3])} servo targets, {len(motors[(frame + 1) % 3::3])} motor speeds")
        if frame == 4:   # three ragdolls go to sleep: every constraint that touches them leaves, with its impulses
            bodies = set().union(*(body_of_ragdoll(r) for r in sleepers))
            island = constraints_of(sim, export, lambda t, b: any(x in bodies for x in b))
            assert len(island) >= 3 * 58
            for h, _, t, _, b, lane, acc in island:
                asleep.append((t, b, lane, acc))
                sim.remove_constraint(h)
            what.append(f"{len(island)} constraints asleep")
        if frame == 9:   # ... and wake up: Solver.Add for each, then the stored impulses written back (the awakener's bulk copy)
            for t, b, lane, acc in asleep:
                h = sim.add_constraint(t, b, lane)
                sim.set_accumulated_impulses(h, acc)
            assert not REQUIRE_RESTORED_IMPULSES or any(np.any(acc != 0) for _, _, _, acc in asleep), "the sleeping joints had accumulated impulses to restore"
            what.append(f"{len(asleep)} constraints awake")
        if frame in (6, 11):  # the LAST ball socket of a type batch removed and another ball socket added: same handle (the pool is LIFO), same type batch, same index
            sockets = constraints_of(sim, export, lambda t, b: t == BALL_SOCKET)
            by_tb = {}
            for c in sockets:
                by_tb.setdefault((c[1], c[2]), []).append(c)
            last = max(next(iter(by_tb.values())), key=lambda c: c[3])
            h, bi, t, i, b, lane, acc = last
            sim.remove_constraint(h)
            other = lane.copy()
            other[0:3] += np.float32(0.05)  # another joint: LocalOffsetA moved (BallSocket.cs:60-65)
            h2 = sim.add_constraint(t, b, other)
            assert h2 == h and sim.constraint_location(h2) == (bi, t, i), "this frame is about a handle that comes back at the same place"
            what.append("a reused handle at the same index")
        if frame == 7:   # a swing limit swapped for an angular motor on the same two bodies: leaves (batch, SwingLimit), arrives in (some batch, AngularMotor)
            limits = constraints_of(sim, export, lambda t, b: t == SWING_LIMIT and b[0] >= 16 * 10)
            h, bi, t, i, b, lane, acc = limits[0]
            sim.remove_constraint(h)
            motor_lane = np.array([0.3, -0.2, 0.1, 3.4e38, 100.0], dtype=np.float32)  # TargetVelocityLocalA, MotorSettings(MaximumForce, Damping)
            h2 = sim.add_constraint(ANGULAR_MOTOR, b, motor_lane)
            assert sim.constraint_location(h2)[0] == bi, "first fit puts the new constraint into the batch the old one freed"
            what.append("a constraint replaced by another type in the same batch")
        sim.validate()
        check_frame(sim, sd, cb, frame, what)
        ops, refreshed, schedule = sim.resident_stats()
        schedules.append(schedule)
        if frame != 7:  # (the angular motor of frame 7 opens a type batch its batch did not have: no plan absorbs that; bepuhip_replan puts the context back a frame later)
            assert schedule == expected_schedule, (frame, what, schedules)
    uploads, _ = sim.timestepper_stats()
    ops, refreshed, schedule = sim.resident_stats()
    assert uploads == 1, "the scene stayed resident: one upload, everything else through the diff and the ranged transfers"
    assert refreshed > 0 and ops > 0
    sim.close()


def test_transfer_rows_matches_the_single_calls_and_registered_memory(hip_solver_factory):
    """bepuhip_transfer_rows_async against the calls it batches: prestep data and impulses of several type batches rewritten by range in one call — from registered host
    memory (read by the kernel itself over the link) and from unregistered memory (staged) — then read back both ways; the solve that follows equals the oracle's on
    the same rewritten scene."""
    import small_scenes
    from bepuphysics2_amd import native
    from bepuphysics2_amd.scene import SolveDescription
    scene = small_scenes.random_graph_scene(5, 900, 4000, [4, 5, 6, 7, 22, 23, 25, 26, 27, 30, 46, 47])
    sd, cb = SolveDescription(2, 3), PoseIntegratorCallbacks()
    for registered in (False, True):
        solver = hip_solver_factory(device=0)
        work = scene.copy()
        if registered:
            # As the reference holds them: every type batch's buffers are sub-allocations of ONE large block (BufferPool.cs:42,83), and the block is what gets registered —
            # hundreds of separately registered heap arrays would share pages with their unregistered neighbours, which no DMA engine forgives.
            tbs = [tb for b in work.batches for tb in b if tb.count]
            floats = sum((a.size + 31) // 32 * 32 for tb in tbs for a in (tb.prestep, tb.accumulated)) + work.bodies.size + 4096
            raw = np.zeros(floats + 1024, dtype=np.float32)
            skip = (-raw.ctypes.data % 4096) // 4
            block = raw[skip:skip + floats]
            at = 0
            def carve(a):
                nonlocal at
                view = block[at:at + a.size].reshape(a.shape)
                view[...] = a
                at += (a.size + 31) // 32 * 32  # 128-byte aligned, like the pool's buffers
                return view
            for tb in tbs:
                tb.prestep, tb.accumulated = carve(tb.prestep), carve(tb.accumulated)
            work.bodies = carve(work.bodies)
            solver.register_host_memory(block)
        solver.upload(work)
        rng = np.random.default_rng(3)
        items = []
        for bi, b in enumerate(work.batches):
            for tb in b:
                if tb.count < 9:
                    continue
                pf, imf = TYPE_TABLE[tb.type_id][1] * 8, TYPE_TABLE[tb.type_id][2] * 8
                bundles = (tb.count + 7) // 8
                first = int(rng.integers(0, bundles - 1))
                n = int(rng.integers(1, bundles - first + 1))
                acc = tb.accumulated.reshape(-1)
                acc[first * imf:(first + n) * imf] *= np.float32(0.5)
                items.append((native.ROWS_UPDATE_IMPULSES, bi, tb.type_id, first, acc[first * imf:(first + n) * imf]))
                if not TYPE_TABLE[tb.type_id][3].startswith("Contact"):
                    continue
                pre = tb.prestep.reshape(-1)
                pre[first * pf:(first + n) * pf] += np.float32(0.001)
                items.append((native.ROWS_UPDATE_PRESTEP, bi, tb.type_id, first, pre[first * pf:(first + n) * pf]))
        assert len(items) > 10
        solver.transfer_rows(items)
        solver.sync()
        # read everything back through the batched call and through the single calls: both must show the host's values
        back = [(native.ROWS_GET_IMPULSES, bi, tb.type_id, 0, np.full(tb.accumulated.size, 7.0, dtype=np.float32)) for bi, b in enumerate(work.batches) for tb in b if tb.count]
        back += [(native.ROWS_GET_PRESTEP, bi, tb.type_id, 0, np.full(tb.prestep.size, 7.0, dtype=np.float32)) for bi, b in enumerate(work.batches) for tb in b if tb.count]
        solver.transfer_rows(back)
        solver.sync()
        for kind, bi, t, _, got in back:
            tb = next(x for x in work.batches[bi] if x.type_id == t)
            lanes = TYPE_TABLE[t][1] if kind == native.ROWS_GET_PRESTEP else TYPE_TABLE[t][2]
            want = (tb.prestep if kind == native.ROWS_GET_PRESTEP else tb.accumulated).reshape(-1, lanes, 8)
            have = got.reshape(-1, lanes, 8)
            live = (np.arange(have.shape[0] * 8) < tb.count).reshape(-1, 8)
            assert np.array_equal(np.where(live[:, None, :], want, 0).view(np.int32), have.view(np.int32)), (registered, kind, bi, t)
        ref = work.copy()
        oracle_ffi.solve(ref, 1 / 60, sd, cb)
        solver.solve(1 / 60, sd, cb, asynchronous=True)
        before = work.bodies.copy()
        solver.get_poses_and_velocities(work.bodies, asynchronous=True)  # (registered: written by a kernel straight into the host's array)
        solver.sync()
        cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
        assert np.array_equal(ref.bodies[:, cols].view(np.int32), work.bodies[:, cols].view(np.int32)), registered
        assert np.array_equal(before[:, 16:].view(np.int32), work.bodies[:, 16:].view(np.int32)), "the inertia half of the host's records is left alone"
        solver.close()
