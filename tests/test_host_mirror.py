"""CPU tests of the host-side mirror: batch colouring (Solver.Add), AOSOA layout, the a2 integration-responsibility prepass
(host C++ vs oracle), SolveDescription semantics and error behaviour."""
import numpy as np
import pytest

import oracle_ffi
import small_scenes
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.scene import (KINEMATIC_MASK, TYPE_TABLE, Scene, SceneBuilder, SolveDescription, from_aosoa, make_body, to_aosoa)


def test_aosoa_round_trip():
    rng = np.random.default_rng(0)
    for count in (0, 1, 7, 8, 9, 100):
        lanes = rng.normal(size=(count, 5)).astype(np.float32)
        for w in (4, 8, 16):
            buf = to_aosoa(lanes, w)
            assert buf.size == ((count + w - 1) // w) * 5 * w
            assert np.array_equal(from_aosoa(buf, count, 5, w), lanes)
            if count:  # BundleIndexing.cs:50-60 addressing
                i, f = count - 1, 3
                assert buf[(i // w) * 5 * w + f * w + (i % w)] == lanes[i, f]


def _random_adds(seed, n_bodies=60, n_constraints=200):
    rng = np.random.default_rng(seed)
    bodies, adds = [], []
    for i in range(n_bodies):
        pos = rng.uniform(-2, 2, 3)
        bodies.append(small_scenes.kinematic_body(rng, pos) if rng.random() < 0.1 else small_scenes.random_dynamic_body(rng, pos))
    kin = [not np.any(b[16:23]) for b in bodies]
    types = sorted(TYPE_TABLE.keys())
    while len(adds) < n_constraints:
        t = int(types[rng.integers(len(types))])
        hs = [int(h) for h in rng.choice(n_bodies, size=TYPE_TABLE[t][0], replace=False)]
        if all(kin[h] for h in hs):
            continue
        adds.append((t, hs, [float(x) for x in small_scenes.prestep_for(rng, t, bodies[hs[0]][4:7], bodies[hs[-1]][4:7])]))
    return bodies, adds


def test_python_and_cpp_solver_add_build_identical_batches():
    bodies, adds = _random_adds(4)
    sb = SceneBuilder()
    sim = HostSimulation.create()
    for b in bodies:
        sb.add_body(b)
        sim.add_body(b[4:7], b[0:4], b[8:11], b[12:15], b[16:22], float(b[22]))
    for t, hs, lane in adds:
        sb.add_constraint(t, hs, lane)
        sim.add_constraint(t, hs, lane)
    sim.validate()
    a, b = sb.build(), sim.export()
    assert len(a.batches) == len(b.batches)
    for ba, bb in zip(a.batches, b.batches):
        assert [tb.type_id for tb in ba] == [tb.type_id for tb in bb]
        for ta, tb in zip(ba, bb):
            assert ta.count == tb.count
            assert np.array_equal(ta.refs_lanes(), tb.refs_lanes())
            assert np.array_equal(ta.prestep_lanes(), tb.prestep_lanes())
    assert np.array_equal(a.bodies[:, :24], b.bodies[:, :24])
    assert sorted(a.constrained_kinematic_handles.tolist()) == sorted(b.constrained_kinematic_handles.tolist())


def test_batch_invariants_and_kinematic_encoding():
    sc = small_scenes.random_graph_scene(12, 150, 500, sorted(TYPE_TABLE.keys()), kinematic_fraction=0.15)
    kin = ~np.any(sc.bodies[:, 16:23] != 0, axis=1)
    for batch in sc.batches:
        seen = set()
        for tb in batch:
            refs = tb.refs_lanes()
            for r in refs.reshape(-1):
                idx = int(r) & 0x3FFFFFFF
                if int(r) & KINEMATIC_MASK:
                    assert kin[idx]
                else:
                    assert not kin[idx]
                    assert idx not in seen, "dynamic body twice in one batch (Solver.cs:1046-1051)"
                    seen.add(idx)
            # trailing lanes of the last bundle are -1 (TypeProcessor.cs:287-298)
            full = tb.body_refs.reshape(-1, tb.bodies, sc.bundle_width)
            pad = tb.count % sc.bundle_width
            if pad:
                assert np.all(full[-1, :, pad:] == -1)


@pytest.mark.parametrize("name,a", [("ragdoll_tube", 40), ("pile", 500), ("pyramid", 2)])
def test_prepass_host_matches_oracle(name, a):
    sim = HostSimulation.scene(name, a, 1, 0, 5)
    sc = sim.export()
    hm, hf, hc = sim.prepare_flags(sc)
    om, of, oc = oracle_ffi.prepare_flags(sc)
    assert np.array_equal(hm, om) and np.array_equal(hf, of) and np.array_equal(hc, oc)
    # merged set == every body referenced by a constraint (dynamic or kinematic)
    referenced = np.zeros(sc.body_count, bool)
    for b in sc.batches:
        for tb in b:
            referenced[tb.refs_lanes().reshape(-1) & 0x3FFFFFFF] = True
    bits = np.unpackbits(om.view(np.uint8), bitorder="little")[: sc.body_count].astype(bool)
    assert np.array_equal(bits[sc.index_to_handle], referenced)


def test_ragdoll_recipe_counts():
    sim = HostSimulation.scene("ragdoll_tube", 10, 1, 0, 5)
    sc = sim.export()
    counts = {}
    for b in sc.batches:
        for tb in b:
            counts[TYPE_TABLE[tb.type_id][3]] = counts.get(TYPE_TABLE[tb.type_id][3], 0) + tb.count
    # RagdollTubeBenchmark.cs:199-498: 11 BallSocket, 15 SwingLimit, 9 TwistLimit, 15 AngularMotor, 2 SwivelHinge, 2 Hinge, 4 TwistServo per ragdoll
    assert counts["BallSocket"] == 110 and counts["SwingLimit"] == 150 and counts["TwistLimit"] == 90 and counts["AngularMotor"] == 150
    assert counts["SwivelHinge"] == 20 and counts["Hinge"] == 20 and counts["TwistServo"] == 40
    assert sc.body_count == 161 and len(sc.constrained_kinematic_handles) == 1
    assert len(sc.batches) >= 16  # the chest carries 4 joints x 4 constraints


def test_solve_description_semantics():
    sd = SolveDescription(4, 1)  # (velocityIterationCount, substepCount) — SolveDescription.cs:55
    assert sd.velocity_iteration_count == 4 and sd.substep_count == 1
    assert SolveDescription(2, 3, velocity_iteration_scheduler=lambda s: [0, 5, -1][s]).iterations().tolist() == [2, 5, 2]
    for bad in ((0, 1), (1, 0)):
        with pytest.raises(ValueError):
            SolveDescription(*bad)
    with pytest.raises(ValueError):
        HostSimulation.create(velocity_iterations=0)


def test_timestep_rejects_nonpositive_dt_and_missing_timestepper():
    sim = HostSimulation.create()
    with pytest.raises(ValueError):
        sim.timestep(0.0)  # Simulation.cs:318-319 ArgumentException
    with pytest.raises(RuntimeError):
        sim.timestep(1 / 60)  # no timestepper attached: this mirror has no CPU solver


def test_rig_scene_keeps_the_graph_and_swaps_the_joint_types():
    """bepuphysics2_amd.synthetic.rig_scene (bench.py's widened_types leg): the ragdoll tube's bodies, batches and constraint counts with seven joint types replaced by
    widened ones of the same body count; the oracle solves it (finite), and the mirror's host removal of a kinematic body's last constraint takes it out of
    Solver.ConstrainedKinematicHandles (Solver.cs:1368-1377)."""
    import oracle_ffi
    from bepuphysics2_amd.hostlib import HostSimulation
    from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks
    from bepuphysics2_amd.synthetic import RIG_REMAP, rig_scene
    sim = HostSimulation.scene("ragdoll_tube", 40, 1, 0, 5)
    plain = sim.export()
    sim.close()
    rigs, sd = rig_scene(40)
    assert rigs.body_count == plain.body_count and rigs.constraint_count == plain.constraint_count and len(rigs.batches) == len(plain.batches)
    for a, b in zip(plain.batches, rigs.batches):
        assert [RIG_REMAP.get(tb.type_id, tb.type_id) for tb in a] == [tb.type_id for tb in b]
        for x, y in zip(a, b):
            assert x.count == y.count and np.array_equal(x.body_refs, y.body_refs) and TYPE_TABLE[x.type_id][0] == TYPE_TABLE[y.type_id][0]
    oracle_ffi.solve(rigs, 1 / 60, sd, PoseIntegratorCallbacks())
    assert np.isfinite(rigs.bodies).all()
    # ConstrainedKinematicHandles follows removals in the C++ host mirror
    sim = HostSimulation.create()
    dyn = sim.add_body((0, 0, 0), (0, 0, 0, 1), (0, 0, 0), (0, 0, 0), (1, 0, 1, 0, 0, 1), 1.0)
    kin = sim.add_body((1, 0, 0), (0, 0, 0, 1), (0, 0, 0), (0, 0, 0.25), (0, 0, 0, 0, 0, 0), 0.0)
    lane = [0.1] * TYPE_TABLE[22][1]
    first = sim.add_constraint(22, [dyn, kin], lane)
    second = sim.add_constraint(22, [dyn, kin], lane)
    assert list(sim.export().constrained_kinematic_indices()) == [kin]
    sim.remove_constraint(first)
    assert list(sim.export().constrained_kinematic_indices()) == [kin]
    sim.remove_constraint(second)
    assert list(sim.export().constrained_kinematic_indices()) == []
    sim.close()
