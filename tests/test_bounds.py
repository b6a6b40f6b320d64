"""PredictBoundingBoxes: behavioural pins of the oracle restatement (CPU) and bit-exact parity of the HIP path through the C ABI (-m gpu).

The oracle text of this stage is derived mechanically from the device text (tools/port_constraints_to_oracle.py), so parity pins the GPU arithmetic and
the CPU tests below pin the transcription: the predicted box must contain the shape at its current pose and at the pose the callback-integrated velocity
leads to, obey the speculative-margin clamp, and the sleep counters must follow PoseIntegrator.UpdateSleepCandidacy."""
import numpy as np
import pytest

import oracle_ffi
import small_scenes
from bepuphysics2_amd.native import (COLLIDABLE_DTYPE, SHAPE_BOX, SHAPE_CAPSULE, SHAPE_CYLINDER, SHAPE_SPHERE, SHAPE_TRIANGLE)
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
    bad["shape_type"][2] = 6  # Compound.Id: stays on the host
    with pytest.raises(native.UnsupportedError):
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
