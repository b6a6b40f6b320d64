"""PredictBoundingBoxes: behavioural pins of the oracle restatement (CPU) and bit-exact parity of the HIP path through the C ABI (-m gpu).

The oracle text of this stage is derived mechanically from the device text (tools/port_constraints_to_oracle.py), so parity pins the GPU arithmetic and
the CPU tests below pin the transcription: the predicted box must contain the shape at its current pose and at the pose the callback-integrated velocity
leads to, obey the speculative-margin clamp, and the sleep counters must follow PoseIntegrator.UpdateSleepCandidacy. Compounds and meshes are the exception: their oracle
text was written from the C# on its own (every child becomes a collidable of the convex path), their device text separately (one lane walks the children)."""
import numpy as np
import pytest

import oracle_ffi
import small_scenes
import wide_ffi
from bepuphysics2_amd.native import (COLLIDABLE_DTYPE, COMPOUND_CHILD_DTYPE, SHAPE_BIG_COMPOUND, SHAPE_BOX, SHAPE_CAPSULE, SHAPE_COMPOUND, SHAPE_CONVEX_HULL, SHAPE_CYLINDER, SHAPE_MESH,
                                      SHAPE_SPHERE, SHAPE_TRIANGLE)
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, make_body

FLOAT_MAX = float(np.finfo(np.float32).max)


def _rotate(q, v):
    x, y, z, w = [float(c) for c in q]
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return r @ np.asarray(v, np.float64)


def _surface_points(shape_type, s, rng, n=400):
    """Points of the shape in its local frame (enough to probe the extent in every direction)."""
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if shape_type == SHAPE_SPHERE:
        return d * s[0]
    if shape_type == SHAPE_CAPSULE:
        return d * s[0] + np.outer(np.sign(d[:, 1]), [0, s[1], 0])
    if shape_type == SHAPE_BOX:
        return np.sign(d) * np.asarray(s[:3])
    if shape_type == SHAPE_CYLINDER:
        radial = d.copy(); radial[:, 1] = 0
        radial /= np.maximum(np.linalg.norm(radial, axis=1, keepdims=True), 1e-9)
        return radial * s[0] + np.outer(np.sign(d[:, 1]), [0, s[1], 0])
    tri = np.asarray(s[:9]).reshape(3, 3)
    w = np.vstack([np.eye(3), rng.dirichlet([1, 1, 1], size=n)])  # the vertices themselves, then interior points
    return w @ tri


def _random_collidables(rng, n, continuous_fraction=0.5):
    c = np.zeros(n, dtype=COLLIDABLE_DTYPE)
    for i in range(n):
        t = int(rng.integers(-1, 5))
        c["shape_type"][i] = t
        if t == SHAPE_SPHERE:
            c["shape"][i, 0] = rng.uniform(0.2, 1.5)
        elif t in (SHAPE_CAPSULE, SHAPE_CYLINDER):
            c["shape"][i, :2] = rng.uniform(0.2, 1.5, 2)
        elif t == SHAPE_BOX:
            c["shape"][i, :3] = rng.uniform(0.2, 1.5, 3)
        elif t == SHAPE_TRIANGLE:
            c["shape"][i, :9] = rng.uniform(-1.5, 1.5, 9)
        c["minimum_speculative_margin"][i] = rng.uniform(0, 0.05)
        c["maximum_speculative_margin"][i] = FLOAT_MAX if rng.random() < 0.5 else rng.uniform(0.05, 0.5)
        c["allow_expansion_beyond_speculative_margin"][i] = int(rng.random() < continuous_fraction)
        c["sleep_threshold"][i] = rng.uniform(0.001, 2.0)
        c["minimum_timesteps_under_threshold"][i] = int(rng.integers(1, 40))
        c["activity"][i] = int(rng.integers(0, 256)) | (int(rng.random() < 0.3) << 8)
    return c


def _random_bodies(rng, n):
    bodies = np.stack([small_scenes.random_dynamic_body(rng, rng.uniform(-5, 5, 3), speed=2.0) if i % 7 else small_scenes.kinematic_body(rng, rng.uniform(-5, 5, 3), angular=(0.3, -1.2, 0.4))
                       for i in range(n)]).astype(np.float32)
    return bodies


def test_predicted_bounds_contain_the_swept_shape():
    rng = np.random.default_rng(21)
    n, dt = 300, 1 / 60
    cb = PoseIntegratorCallbacks()
    bodies, coll = _random_bodies(rng, n), _random_collidables(rng, n, continuous_fraction=1.0)
    coll["maximum_speculative_margin"] = FLOAT_MAX
    before = bodies.copy()
    out = oracle_ffi.predict_bounding_boxes(bodies, dt, cb, coll)
    assert np.array_equal(bodies, before)  # the integrated velocity is used for the prediction only (:331-333)
    damp_l, damp_a = (1 - cb.linear_damping) ** dt, (1 - cb.angular_damping) ** dt
    for i in range(n):
        t = int(coll["shape_type"][i])
        if t < 0:
            assert not out["min"][i].any() and not out["max"][i].any()
            continue
        pts = _surface_points(t, coll["shape"][i], rng)
        pos, q = bodies[i, 4:7].astype(np.float64), bodies[i, 0:4]
        kinematic = not bodies[i, 16:23].any()
        lin = bodies[i, 8:11].astype(np.float64) if kinematic else (bodies[i, 8:11] + np.asarray(cb.gravity) * dt) * damp_l
        ang = bodies[i, 12:15].astype(np.float64) if kinematic else bodies[i, 12:15] * damp_a
        now = np.array([pos + _rotate(q, p) for p in pts])
        # end pose: translate by v dt, rotate by |w| dt about w (Rodrigues)
        angle = np.linalg.norm(ang) * dt
        axis = ang / max(np.linalg.norm(ang), 1e-12)
        rel = now - pos
        rot = rel * np.cos(angle) + np.cross(axis, rel) * np.sin(angle) + np.outer(rel @ axis, axis) * (1 - np.cos(angle))
        later = pos + lin * dt + rot
        lo, hi = out["min"][i].astype(np.float64) - 1e-4, out["max"][i].astype(np.float64) + 1e-4
        assert (now >= lo).all() and (now <= hi).all(), (i, t)
        assert (later >= lo).all() and (later <= hi).all(), (i, t)
        # and it is not absurdly loose: within the shape's radius plus the displacement of a tight box
        extent = np.abs(rel).max() + np.linalg.norm(lin) * dt + np.linalg.norm(rel, axis=1).max() * angle + 1e-3
        assert (hi - pos <= 1.05 * extent + 0.02).all() and (pos - lo <= 1.05 * extent + 0.02).all(), (i, t)  # (sampled surfaces miss the exact extreme points)


def test_speculative_margin_clamps_and_discrete_expansion():
    rng = np.random.default_rng(22)
    body = make_body(position=(1, 2, 3), linear=(30.0, 0, 0))  # 0.5 units per frame: far beyond the margins below
    body[16:23] = (1, 0, 1, 0, 0, 1, 1)
    cb = PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0.0, angular_damping=0.0)
    c = np.zeros(2, dtype=COLLIDABLE_DTYPE)
    c["shape_type"] = SHAPE_SPHERE
    c["shape"][:, 0] = 0.5
    c["minimum_speculative_margin"] = 0.0
    c["maximum_speculative_margin"] = 0.1
    c["allow_expansion_beyond_speculative_margin"] = (0, 1)  # Discrete vs Passive/Continuous (Collidable.cs:59)
    c["sleep_threshold"] = 0.01
    c["minimum_timesteps_under_threshold"] = 32
    out = oracle_ffi.predict_bounding_boxes(np.stack([body, body]), 1 / 60, cb, c)
    assert np.allclose(out["speculative_margin"], 0.1)                       # min(maximum, |v| dt)
    assert np.isclose(out["max"][0, 0], 1 + 0.5 + 0.1) and np.isclose(out["max"][1, 0], 1 + 0.5 + 0.5)   # discrete: the box grows by the margin only
    assert np.isclose(out["min"][0, 0], 0.5) and np.isclose(out["min"][1, 0], 0.5)


def test_sleep_candidacy_counters():
    cb = PoseIntegratorCallbacks()
    slow, fast = make_body(linear=(0.01, 0, 0)), make_body(linear=(2, 0, 0))
    c = np.zeros(4, dtype=COLLIDABLE_DTYPE)
    c["shape_type"] = -1
    c["sleep_threshold"] = 0.01
    c["minimum_timesteps_under_threshold"] = 3
    c["activity"] = (1, 2, 255 | 0x100, 7 | 0x100)
    out = oracle_ffi.predict_bounding_boxes(np.stack([slow, slow, slow, fast]), 1 / 60, cb, c)
    assert list(out["activity"]) == [2, 3 | 0x100, 255 | 0x100, 0]  # below threshold: count up (saturating), candidate from the minimum on; above: reset


@pytest.mark.gpu
def test_hip_predict_bounding_boxes_matches_the_oracle(hip_solver_factory):
    rng = np.random.default_rng(23)
    n = 5000
    bodies, coll = _random_bodies(rng, n), _random_collidables(rng, n)
    solver = hip_solver_factory()
    solver.set_bodies(bodies)
    for cb in (PoseIntegratorCallbacks(), PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True)):
        want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll)
        got = solver.predict_bounding_boxes(1 / 60, cb, coll)
        assert np.array_equal(want.view(np.int32), got.view(np.int32))
    assert np.array_equal(solver.get_bodies(n).view(np.int32)[:, :15], bodies.view(np.int32)[:, :15])  # bodies untouched
    # device-resident records: shapes stay on the device, the sleep counters carry over from call to call exactly as the oracle's do when fed its own output
    cb = PoseIntegratorCallbacks()
    solver.set_collidables(coll)
    chained = coll.copy()
    for _ in range(3):
        want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, chained)
        got = solver.predict_bounding_boxes(1 / 60, cb)
        assert np.array_equal(want.view(np.int32), got.view(np.int32))
        chained["activity"] = want["activity"]
    from bepuphysics2_amd import native
    bad = coll[:4].copy()
    bad["shape_type"][2] = 9  # not a shape type of the library
    with pytest.raises(native.UnsupportedError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks(), bad)
    bad["shape_type"][2] = 6  # Compound.Id without a compound table: not a valid index
    with pytest.raises(ValueError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks(), bad)
    bad["shape_type"][2] = 5  # ConvexHull.Id without a hull table: not a valid hull index
    with pytest.raises(ValueError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks(), bad)
    with pytest.raises(ValueError):
        solver.predict_bounding_boxes(0.0, PoseIntegratorCallbacks(), coll[:4])


def _random_hulls(rng, count):
    """Point clouds standing in for ConvexHull.Points (the bounds only look at the points): 4 to 40 points each, off-centre, different sizes."""
    return [(rng.normal(size=(int(rng.integers(4, 41)), 3)) * rng.uniform(0.2, 1.5, 3) + rng.uniform(-0.3, 0.3, 3)).astype(np.float32) for _ in range(count)]


def _hull_collidables(rng, n, hull_count):
    c = _random_collidables(rng, n)
    for i in range(n):
        if i % 3 == 0:
            c["shape_type"][i] = 5  # ConvexHull.Id
            c["shape"][i] = 0
            c["shape"][i, 0] = float(rng.integers(hull_count))
    return c


def test_convex_hull_bounds_contain_every_rotated_point():
    """ConvexHullWide.GetBounds (ConvexHull.cs:319-364): the box holds every point of the hull at the body's orientation and touches the extreme ones."""
    rng = np.random.default_rng(41)
    hulls = _random_hulls(rng, 12)
    n = 300
    bodies = _random_bodies(rng, n)
    bodies[:, 8:11] = 0
    bodies[:, 12:15] = 0  # at rest: no linear or angular expansion
    coll = _hull_collidables(rng, n, len(hulls))
    coll["minimum_speculative_margin"] = 0
    cb = PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0, angular_damping=0)
    out = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls)
    checked = 0
    for i in range(n):
        if coll["shape_type"][i] != 5:
            continue
        pts = hulls[int(coll["shape"][i, 0])]
        world = np.stack([_rotate(bodies[i, 0:4], p) for p in pts])
        lo, hi = world.min(axis=0), world.max(axis=0)
        assert np.allclose(out["min"][i], lo + bodies[i, 4:7], atol=2e-5) and np.allclose(out["max"][i], hi + bodies[i, 4:7], atol=2e-5), i
        checked += 1
    assert checked > 50


@pytest.mark.gpu
def test_hip_convex_hull_bounds_match_the_oracle(hip_solver_factory):
    rng = np.random.default_rng(43)
    hulls = _random_hulls(rng, 64)
    n = 4000
    bodies, coll = _random_bodies(rng, n), _hull_collidables(rng, n, len(hulls))
    solver = hip_solver_factory()
    solver.set_bodies(bodies)
    solver.set_convex_hulls(hulls)
    for cb in (PoseIntegratorCallbacks(), PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True)):
        want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls)
        got = solver.predict_bounding_boxes(1 / 60, cb, coll)
        assert np.array_equal(want.view(np.int32), got.view(np.int32))
    solver.set_collidables(coll)  # resident records + resident hull table
    got = solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks())
    assert np.array_equal(oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, PoseIntegratorCallbacks(), coll, hulls).view(np.int32), got.view(np.int32))
    bad = coll[:6].copy()
    bad["shape_type"][1] = 5
    bad["shape"][1, 0] = len(hulls)  # one past the table
    with pytest.raises(ValueError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks(), bad)


# ---- compounds and meshes (BoundingBoxBatcher.ExecuteCompoundBatch / ExecuteHomogeneousCompoundBatch) ----
def _random_quaternion(rng):
    q = rng.normal(size=4)
    return (q / np.linalg.norm(q)).astype(np.float32)


def _random_compounds(rng, count, hull_count):
    """Children of every convex type (hulls included), 1 to 9 per compound, scattered around the compound's origin."""
    compounds = []
    for _ in range(count):
        kids = np.zeros(int(rng.integers(1, 10)), dtype=COMPOUND_CHILD_DTYPE)
        for k in range(kids.shape[0]):
            t = int(rng.integers(0, 6))
            kids["shape_type"][k] = t
            if t == SHAPE_SPHERE:
                kids["shape"][k, 0] = rng.uniform(0.1, 0.8)
            elif t in (SHAPE_CAPSULE, SHAPE_CYLINDER):
                kids["shape"][k, :2] = rng.uniform(0.1, 0.8, 2)
            elif t == SHAPE_BOX:
                kids["shape"][k, :3] = rng.uniform(0.1, 0.8, 3)
            elif t == SHAPE_TRIANGLE:
                kids["shape"][k, :9] = rng.uniform(-0.8, 0.8, 9)
            else:
                kids["shape"][k, 0] = float(rng.integers(hull_count))
            kids["local_position"][k] = rng.uniform(-2, 2, 3)
            kids["local_orientation"][k] = _random_quaternion(rng)
        compounds.append(kids)
    return compounds


def _random_meshes(rng, count):
    return [((rng.normal(size=(int(rng.integers(1, 60)), 3, 3)) * rng.uniform(0.3, 2.0)).astype(np.float32), rng.uniform(0.5, 2.0, 3).astype(np.float32)) for _ in range(count)]


def _every_shape_collidables(rng, n, hull_count, compound_count, mesh_count):
    c = _hull_collidables(rng, n, hull_count)
    for i in range(n):
        r = i % 7
        if r in (1, 2):
            c["shape_type"][i] = SHAPE_COMPOUND if r == 1 else SHAPE_BIG_COMPOUND
            c["shape"][i] = 0
            c["shape"][i, 0] = float(rng.integers(compound_count))
        elif r == 4:
            c["shape_type"][i] = SHAPE_MESH
            c["shape"][i] = 0
            c["shape"][i, 0] = float(rng.integers(mesh_count))
    return c


def _spinning_bodies(rng, n):
    """Some bodies spin fast enough that a compound child's angular share exceeds its offset (the capped branch of Compound.cs:213-216)."""
    bodies = _random_bodies(rng, n)
    bodies[::3, 12:15] *= 40
    return bodies


def test_compound_bounds_are_the_union_of_the_children_as_bodies_of_their_own():
    """Compound.AddChildBoundsToBatcher: every child is bounded like a convex body at pose (parent x local) with the parent's angular velocity and the linear
    velocity its offset picks up; the compound's box is the union, its margin the largest child margin. Checked against the oracle's own convex path fed by hand."""
    rng = np.random.default_rng(51)
    hulls = _random_hulls(rng, 8)
    compounds = _random_compounds(rng, 20, len(hulls))
    n = 120
    bodies = _spinning_bodies(rng, n)
    coll = _random_collidables(rng, n)
    coll["shape_type"] = SHAPE_COMPOUND
    coll["shape"] = 0
    coll["shape"][:, 0] = rng.integers(len(compounds), size=n)
    cb = PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0, angular_damping=0)  # velocity callback = identity: the child velocities below are the stored ones
    out = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls, compounds)
    capped = 0
    for i in range(n):
        kids = compounds[int(coll["shape"][i, 0])]
        q, pos, lin, ang = bodies[i, 0:4].astype(np.float64), bodies[i, 4:7].astype(np.float64), bodies[i, 8:11].astype(np.float64), bodies[i, 12:15].astype(np.float64)
        lo, hi, margin = np.full(3, np.inf), np.full(3, -np.inf), 0.0
        for kid in kids:
            offset = _rotate(q, kid["local_position"])
            swing = np.cross(ang, offset)
            if swing @ swing > offset @ offset:
                swing *= np.sqrt(offset @ offset) / np.sqrt(swing @ swing)
                capped += 1
            lx, ly, lz, lw = [float(v) for v in kid["local_orientation"]]
            px, py, pz, pw = q
            child_q = np.array([lw * px + lx * pw + lz * py - ly * pz, lw * py + ly * pw + lx * pz - lz * px, lw * pz + lz * pw + ly * px - lx * py, lw * pw - lx * px - ly * py - lz * pz])
            one = np.zeros((1, 32), np.float32)
            one[0, 0:4], one[0, 4:7], one[0, 8:11], one[0, 12:15] = child_q, offset + pos, lin + swing, ang
            one[0, 16:23] = bodies[i, 16:23]
            as_body = coll[i:i + 1].copy()
            as_body["shape_type"], as_body["shape"] = kid["shape_type"], kid["shape"]
            child = oracle_ffi.predict_bounding_boxes(one, 1 / 60, cb, as_body, hulls)
            lo, hi, margin = np.minimum(lo, child["min"][0]), np.maximum(hi, child["max"][0]), max(margin, float(child["speculative_margin"][0]))
        assert np.allclose(out["min"][i], lo, rtol=1e-5, atol=1e-5) and np.allclose(out["max"][i], hi, rtol=1e-5, atol=1e-5), i
        assert np.isclose(out["speculative_margin"][i], margin, rtol=1e-5, atol=1e-6), i
    assert capped > 20


def test_mesh_bounds_contain_every_rotated_vertex_and_follow_the_box_heuristic():
    """Mesh.ComputeBounds (Mesh.cs:232-255) + ExecuteHomogeneousCompoundBatch (:225-266): at rest the box is the extent of the scaled, rotated vertices; moving, it grows
    by the linear sweep and by an angular expansion bounded by (largest corner distance - smallest face distance) of that box."""
    rng = np.random.default_rng(53)
    meshes = _random_meshes(rng, 10)
    n = 200
    bodies = _random_bodies(rng, n)
    coll = _random_collidables(rng, n)
    coll["shape_type"], coll["shape"] = SHAPE_MESH, 0
    coll["shape"][:, 0] = rng.integers(len(meshes), size=n)
    coll["minimum_speculative_margin"] = 0
    coll["allow_expansion_beyond_speculative_margin"] = 1
    cb = PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0, angular_damping=0)
    at_rest = bodies.copy()
    at_rest[:, 8:11] = 0
    at_rest[:, 12:15] = 0
    still = oracle_ffi.predict_bounding_boxes(at_rest, 1 / 60, cb, coll, meshes=meshes)
    moving = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, meshes=meshes)
    for i in range(n):
        tris, scale = meshes[int(coll["shape"][i, 0])]
        world = np.stack([_rotate(bodies[i, 0:4], v * scale) for v in tris.reshape(-1, 3)])
        lo, hi = world.min(axis=0), world.max(axis=0)
        assert np.allclose(still["min"][i], lo + bodies[i, 4:7], atol=5e-5) and np.allclose(still["max"][i], hi + bodies[i, 4:7], atol=5e-5), i
        assert still["speculative_margin"][i] == 0
        sweep = bodies[i, 8:11].astype(np.float64) / 60
        corner = np.linalg.norm(np.maximum(np.abs(lo), np.abs(hi)))
        cap = corner - np.min(np.minimum(np.abs(lo), np.abs(hi)))
        grow_lo, grow_hi = (still["min"][i] - moving["min"][i]).astype(np.float64), (moving["max"][i] - still["max"][i]).astype(np.float64)
        angular = grow_hi - np.maximum(sweep, 0)
        assert np.all(grow_lo > -1e-5) and np.all(grow_hi > -1e-5)
        assert np.allclose(angular, angular[0], atol=1e-4) and angular[0] <= cap + 1e-4, i  # one scalar angular expansion on every axis, never above the cap
        assert np.allclose(grow_lo, angular[0] - np.minimum(sweep, 0), atol=1e-4), i


@pytest.mark.gpu
def test_hip_compound_and_mesh_bounds_match_the_oracle(hip_solver_factory):
    """Every shape type the reference registers in one body set (primitives, hulls, compounds with hull children, big compounds, meshes), fast spinners included: bit-exact."""
    rng = np.random.default_rng(57)
    hulls, meshes = _random_hulls(rng, 32), _random_meshes(rng, 24)
    compounds = _random_compounds(rng, 80, len(hulls))
    n = 6000
    bodies = _spinning_bodies(rng, n)
    coll = _every_shape_collidables(rng, n, len(hulls), len(compounds), len(meshes))
    assert set(np.unique(coll["shape_type"])) >= {0, 1, 2, 3, 4, 5, 6, 7, 8}
    solver = hip_solver_factory()
    solver.set_bodies(bodies)
    solver.set_convex_hulls(hulls)
    solver.set_compounds(compounds)
    solver.set_meshes(meshes)
    for cb in (PoseIntegratorCallbacks(), PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True)):
        want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls, compounds, meshes)
        got = solver.predict_bounding_boxes(1 / 60, cb, coll)
        assert np.array_equal(want.view(np.int32), got.view(np.int32))
    solver.set_collidables(coll)  # resident records and tables; the sleep counters chain
    chained = coll.copy()
    for _ in range(2):
        want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, PoseIntegratorCallbacks(), chained, hulls, compounds, meshes)
        got = solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks())
        assert np.array_equal(want.view(np.int32), got.view(np.int32))
        chained["activity"] = want["activity"]
    # the boundary refuses what the tables do not hold
    bad = coll[:8].copy()
    bad["shape_type"][1], bad["shape"][1, 0] = SHAPE_MESH, len(meshes)
    with pytest.raises(ValueError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks(), bad)
    nested = np.zeros(1, dtype=COMPOUND_CHILD_DTYPE)
    nested["shape_type"] = SHAPE_COMPOUND  # children are convex
    with pytest.raises(ValueError):
        solver.set_compounds([nested])
    solver.set_convex_hulls(hulls[:2])  # compounds name hulls the smaller table no longer has
    with pytest.raises(ValueError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks(), coll)
    from bepuphysics2_amd import native
    with pytest.raises(native.BepuHipError):
        solver.predict_bounding_boxes(1 / 60, PoseIntegratorCallbacks())


def test_velocity_callback_runs_on_whole_bundles_as_the_reference_writes_it():
    """PoseIntegrator.cs:337-338 calls the callback on a bundle as soon as ONE lane is to be integrated and never masks the result; DemoCallbacks.cs:99-109 ignores
    the mask. So a kinematic body's predicted box feels gravity exactly when a dynamic body shares its bundle of Vector<float>.Count bodies — as written, per bundle width."""
    rng = np.random.default_rng(61)
    n = 32
    bodies = np.stack([small_scenes.kinematic_body(rng, rng.uniform(-5, 5, 3), angular=(0, 0, 0)) for _ in range(n)]).astype(np.float32)
    bodies[:, 8:11] = 0
    bodies[5] = small_scenes.random_dynamic_body(rng, (0, 0, 0), speed=0.0)  # the only dynamic body: bundle 0 for widths 8 and 16, bundle 1 for width 4
    coll = np.zeros(n, dtype=COLLIDABLE_DTYPE)
    coll["shape_type"], coll["shape"][:, 0] = SHAPE_SPHERE, 0.5
    coll["maximum_speculative_margin"], coll["allow_expansion_beyond_speculative_margin"] = FLOAT_MAX, 1
    cb = PoseIntegratorCallbacks(gravity=(0, -10, 0), linear_damping=0, angular_damping=0)
    for width in (4, 8, 16):
        out = oracle_ffi.predict_bounding_boxes(bodies, 0.1, cb, coll, bundle_width=width)
        swept = np.isclose(out["max"][:, 1] - out["min"][:, 1], 1.0 + 10 * 0.1 * 0.1, atol=1e-5)  # diameter + |g| dt^2 of downward sweep
        still = np.isclose(out["max"][:, 1] - out["min"][:, 1], 1.0, atol=1e-6)
        in_bundle = (np.arange(n) // width) == (5 // width)
        assert np.array_equal(swept, in_bundle) and np.array_equal(still, ~in_bundle), width
    everyone = oracle_ffi.predict_bounding_boxes(bodies, 0.1, PoseIntegratorCallbacks(gravity=(0, -10, 0), linear_damping=0, angular_damping=0, integrate_velocity_for_kinematics=True), coll)
    assert np.allclose(everyone["max"][:, 1] - everyone["min"][:, 1], 1.1, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("width", [4, 8, 16])
def test_hip_bundle_wide_velocity_callback_matches_the_oracle(hip_solver_factory, width):
    """Kinematic bodies scattered among dynamic ones, runs of kinematic bodies longer than a bundle, a ragged last bundle: the device reads the bundle's 'any lane integrates'
    from a wave ballot, the oracle walks bundles like the reference."""
    rng = np.random.default_rng(63 + width)
    n = 4099
    bodies = _random_bodies(rng, n)
    for start in rng.integers(0, n - 40, size=30):  # runs of kinematic bodies that cover whole bundles
        for i in range(int(start), int(start) + int(rng.integers(3, 40))):
            bodies[i] = small_scenes.kinematic_body(rng, rng.uniform(-5, 5, 3), angular=(0.3, -1.2, 0.4))
    coll = _random_collidables(rng, n)
    solver = hip_solver_factory(bundle_width=width)
    solver.set_bodies(bodies)
    cb = PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2)
    want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, bundle_width=width)
    got = solver.predict_bounding_boxes(1 / 60, cb, coll)
    assert np.array_equal(want.view(np.int32), got.view(np.int32))
    per_body = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, bundle_width=4 if width != 4 else 16)
    assert not np.array_equal(want.view(np.int32), per_body.view(np.int32))  # the bundle width is visible in the result


def test_the_two_restatements_of_predict_bounding_boxes_agree_bit_for_bit():
    """oracle/bepu_bounds.h (one body at a time, its convex text shared with the device) against oracle/wide/wide_bounds.h (the reference's own shape, from the C# alone:
    bundles of eight bodies, the batcher with its per-type flushes of sixteen, TShapeWide.GetBounds, merge continuations for compound children, the scalar mesh path)."""
    rng = np.random.default_rng(71)
    hulls, meshes = _random_hulls(rng, 24), _random_meshes(rng, 12)
    compounds = _random_compounds(rng, 40, len(hulls))
    n = 3003  # a ragged last bundle
    bodies = _spinning_bodies(rng, n)
    for start in rng.integers(0, n - 40, size=20):  # all-kinematic bundles among the mixed ones
        for i in range(int(start), int(start) + int(rng.integers(3, 30))):
            bodies[i] = small_scenes.kinematic_body(rng, rng.uniform(-5, 5, 3), angular=(0.3, -1.2, 0.4))
    coll = _every_shape_collidables(rng, n, len(hulls), len(compounds), len(meshes))
    assert set(np.unique(coll["shape_type"])) == {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8}
    for cb in (PoseIntegratorCallbacks(), PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True)):
        chained_a, chained_b = coll.copy(), coll.copy()
        for _ in range(3):  # the sleep counters chain identically
            a = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, chained_a, hulls, compounds, meshes)
            b = wide_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, chained_b, hulls, compounds, meshes)
            assert np.array_equal(a.view(np.int32), b.view(np.int32)), np.flatnonzero((a.view(np.int32).reshape(n, -1) != b.view(np.int32).reshape(n, -1)).any(axis=1))[:10]
            chained_a["activity"], chained_b["activity"] = a["activity"], b["activity"]


@pytest.mark.gpu
def test_hip_wave_per_body_pass_matches_the_oracle_on_large_shapes_and_on_ties(hip_solver_factory, monkeypatch):
    """Compounds, meshes and hulls with more entries than a wave has lanes (several strides of the wave-per-body pass), and axis-aligned bodies whose rotated
    coordinates are exact zeros of both signs: which of two equal candidates a minimum keeps decides the sign of a zero, and the wave merge must keep the one the
    serial reading keeps. The one-lane-per-body path (BEPUHIP_BOUNDS_ONE_LANE) must give the same bits."""
    rng = np.random.default_rng(67)
    hulls = _random_hulls(rng, 6) + [(rng.normal(size=(k, 3)) * 1.3).astype(np.float32) for k in (49, 64, 65, 300)]
    big = np.zeros(700, dtype=COMPOUND_CHILD_DTYPE)
    big["shape_type"] = rng.integers(0, 5, size=700)
    big["shape"][:, :9] = rng.uniform(0.1, 0.6, (700, 9))
    big["local_position"] = rng.uniform(-6, 6, (700, 3))
    big["local_orientation"] = np.stack([_random_quaternion(rng) for _ in range(700)])
    flat = np.zeros(70, dtype=COMPOUND_CHILD_DTYPE)  # boxes on a grid in the plane y = 0, unrotated: child boxes share faces, so maxima tie exactly
    flat["shape_type"], flat["shape"][:, :3] = SHAPE_BOX, 0.5
    flat["local_position"][:, 0], flat["local_position"][:, 2] = np.arange(70) % 10 - 4.5, np.arange(70) // 10 - 3.0
    flat["local_orientation"][:, 3] = 1
    compounds = _random_compounds(rng, 12, len(hulls)) + [big, flat]
    grid = np.array([[[x, 0, z], [x + 1, 0, z], [x, 0, z + 1]] for x in range(-8, 8) for z in range(-8, 8)], np.float32)  # 256 coplanar triangles through the origin
    meshes = _random_meshes(rng, 6) + [((rng.normal(size=(2000, 3, 3))).astype(np.float32), np.ones(3, np.float32)), (grid, np.array([1, 1, 1], np.float32)),
                                       (grid, np.array([-1, 1, -0.5], np.float32))]
    n = 900
    bodies = _spinning_bodies(rng, n)
    axis_aligned = [(0, 0, 0, 1), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0.70710677, 0, 0, 0.70710677), (0, -0.70710677, 0, 0.70710677)]
    for i in range(0, n, 2):
        bodies[i, 0:4] = axis_aligned[(i // 2) % len(axis_aligned)]
        if i % 4 == 0:
            bodies[i, 4:7] = 0
            bodies[i, 8:11] = 0
            bodies[i, 12:15] = 0
    coll = _every_shape_collidables(rng, n, len(hulls), len(compounds), len(meshes))
    for i in range(n):  # make sure the special entries are used, on axis-aligned and on tumbling bodies
        t = coll["shape_type"][i]
        if t == SHAPE_CONVEX_HULL:
            coll["shape"][i, 0] = len(hulls) - 1 - (i % 4)
        elif t in (SHAPE_COMPOUND, SHAPE_BIG_COMPOUND) and i % 3:
            coll["shape"][i, 0] = len(compounds) - 1 - (i % 2)
        elif t == SHAPE_MESH and i % 3:
            coll["shape"][i, 0] = len(meshes) - 1 - (i % 3)
    cb = PoseIntegratorCallbacks(gravity=(0, -9, 0), linear_damping=0.1, angular_damping=0.2)
    want = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls, compounds, meshes)
    assert (want.view(np.int32).reshape(n, 8)[:, [0, 1, 2, 4, 5, 6]] == 0).any()  # exact zeros reach the result (the final position + (box + expansion) turns -0 into +0 unless all three are -0)
    for one_lane in (False, True):
        if one_lane:
            monkeypatch.setenv("BEPUHIP_BOUNDS_ONE_LANE", "1")
        solver = hip_solver_factory()
        solver.set_bodies(bodies)
        solver.set_convex_hulls(hulls)
        solver.set_compounds(compounds)
        solver.set_meshes(meshes)
        got = solver.predict_bounding_boxes(1 / 60, cb, coll)
        bad = np.flatnonzero((want.view(np.int32).reshape(n, 8) != got.view(np.int32).reshape(n, 8)).any(axis=1))
        assert bad.size == 0, (one_lane, bad[:10], coll["shape_type"][bad[:10]], want[bad[:3]], got[bad[:3]])


def test_both_restatements_reproduce_the_committed_bounds_fixture():
    """tests/golden/bounds.npz (tests/golden/make_golden.py): a regression pin of the stage on all nine shape types — the reference holds no vectors for it."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    bodies, coll, hulls, compounds, meshes = make_golden.bounds_inputs()
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bounds.npz"))
    for name, cb in (("default", PoseIntegratorCallbacks()),
                     ("kinematics_integrated", PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True))):
        for restatement in (oracle_ffi, wide_ffi):
            got = restatement.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls, compounds, meshes)
            assert np.array_equal(got.view(np.int32).reshape(-1, 8), golden[name]), (name, restatement.__name__)


@pytest.mark.gpu
def test_hip_reproduces_the_committed_bounds_fixture(hip_solver_factory):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    bodies, coll, hulls, compounds, meshes = make_golden.bounds_inputs()
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bounds.npz"))
    solver = hip_solver_factory()
    solver.set_bodies(bodies)
    solver.set_convex_hulls(hulls)
    solver.set_compounds(compounds)
    solver.set_meshes(meshes)
    for name, cb in (("default", PoseIntegratorCallbacks()),
                     ("kinematics_integrated", PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True))):
        got = solver.predict_bounding_boxes(1 / 60, cb, coll)
        assert np.array_equal(got.view(np.int32).reshape(-1, 8), golden[name]), name
