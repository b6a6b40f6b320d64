"""CPU tests of the oracle: the reference's documented error bounds for its trig approximations, the reference's own
microbenchmark inputs (analytic zero answers), committed golden vectors, and physical invariants that catch sign/transcription
errors in the jacobians (the oracle's parity with the C# is otherwise unpinned — SURVEY.md §8c)."""
import os

import numpy as np
import pytest

import oracle_ffi
import small_scenes
from bepuphysics2_amd.scene import HOT_PATH_TYPES, TYPE_TABLE, WIDENED_TYPES, PoseIntegratorCallbacks, SolveDescription, make_body

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_trig_error_bounds_documented_by_reference():
    # BepuUtilities/MathHelper.cs:270,306,348: "Maximum error a little below 8e-7 (cos) / 5e-7 (sin) for -2pi..2pi", acos "< 5.17e-07".
    x = np.linspace(-2 * np.pi, 2 * np.pi, 200001).astype(np.float32)
    s, c, _ = oracle_ffi.math_probe(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 5.2e-7 + 2e-7  # + fp32 range-reduction noise the reference mentions
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 8e-7 + 2e-7
    u = np.linspace(-1, 1, 100001).astype(np.float32)
    _, _, a = oracle_ffi.math_probe(u)
    assert np.abs(a - np.arccos(u.astype(np.float64))).max() < 5.17e-7 + 3e-7
    # clamping outside [-1, 1]
    _, _, a = oracle_ffi.math_probe(np.asarray([-1.5, 1.5], np.float32))
    assert a[0] == np.float32(np.pi) and a[1] == 0


def test_type_table_matches_oracle():
    import ctypes as C
    lib = oracle_ffi.load()
    for tid, (nb, pf, imf, name) in TYPE_TABLE.items():
        b, p, i, inc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.oracle_type_info(tid, C.byref(b), C.byref(p), C.byref(i), C.byref(inc)) == 0
        assert (b.value, p.value, i.value) == (nb, pf, imf), name
        assert bool(inc.value) == name.startswith("Contact")


def test_reference_microbenchmark_inputs_give_exact_zero():
    """TwoBodyConstraintBenchmarks.cs:42-117: zero velocities, zero depth/error => every velocity and impulse stays exactly 0."""
    import golden.make_golden as mg
    for name, type_id, lane in mg.microbench_inputs():
        a, b, acc = mg.run_micro(type_id, lane)
        assert not np.any(a[8:15]) and not np.any(b[8:15]) and not np.any(acc), name


def test_golden_vectors_regression():
    import golden.make_golden as mg
    g = np.load(os.path.join(GOLDEN, "microbench.npz"))
    va = np.asarray([0.3, -0.2, 0.1, 0.05, 0.4, -0.3], np.float32)
    vb = np.asarray([-0.1, 0.25, 0.0, -0.2, 0.1, 0.15], np.float32)
    for name, type_id, lane in mg.microbench_inputs():
        a, b, acc = mg.run_micro(type_id, lane, va, vb, iterations=8)
        assert np.array_equal(a.view(np.int32), g[f"microv_{name}_a"].view(np.int32))
        assert np.array_equal(b.view(np.int32), g[f"microv_{name}_b"].view(np.int32))
        assert np.array_equal(acc.view(np.int32), g[f"microv_{name}_acc"].view(np.int32))
    s = np.load(os.path.join(GOLDEN, "small_scenes.npz"))
    sd, cb = SolveDescription(2, 8), PoseIntegratorCallbacks()
    for seed, types in ((1, HOT_PATH_TYPES), (2, [0, 1, 2, 3, 4, 5, 6, 7]), (3, [22, 23, 25, 26, 27, 30, 46, 47])):
        sc = small_scenes.random_graph_scene(seed, 120, 300, types)
        for _ in range(2):
            oracle_ffi.solve(sc, 1 / 60, sd, cb)
        assert np.array_equal(sc.bodies.view(np.int32), s[f"graph{seed}_bodies"].view(np.int32))
    w = np.load(os.path.join(GOLDEN, "widened_types.npz"))  # SURVEY 8(f) types, pinned the same way as they are added
    for type_id in WIDENED_TYPES:
        sc = small_scenes.random_graph_scene(400 + type_id, 120, 300, [type_id])
        for _ in range(2):
            oracle_ffi.solve(sc, 1 / 60, sd, cb)
        assert np.array_equal(sc.bodies.view(np.int32), w[f"type{type_id}_bodies"].view(np.int32))


def _momentum(bodies):
    """Linear and angular momentum (about the origin) of dynamic bodies, using the WORLD inverse inertia slots."""
    p_tot, l_tot = np.zeros(3), np.zeros(3)
    for b in bodies.astype(np.float64):
        if b[30] == 0:
            continue
        m = 1.0 / b[30]
        inv = np.array([[b[24], b[25], b[27]], [b[25], b[26], b[28]], [b[27], b[28], b[29]]])
        inertia = np.linalg.inv(inv)
        v, w, r = b[8:11], b[12:15], b[4:7]
        p_tot += m * v
        l_tot += np.cross(r, m * v) + inertia @ w
    return p_tot, l_tot


@pytest.mark.parametrize("type_id", [t for t, v in TYPE_TABLE.items() if v[0] == 2])
def test_two_body_constraints_conserve_momentum(type_id):
    """Each two-body constraint applies equal and opposite impulses (at a common point for contacts / ball sockets):
    linear and angular momentum of an isolated dynamic pair are conserved by WarmStart + Solve."""
    rng = np.random.default_rng(40 + type_id)
    pa, pb = rng.uniform(-1, 1, 3).astype(np.float32), rng.uniform(-1, 1, 3).astype(np.float32)
    a, b = small_scenes.random_dynamic_body(rng, pa), small_scenes.random_dynamic_body(rng, pb)
    name = TYPE_TABLE[type_id][3]
    for body in (a, b):  # world inertia = R^T I_local R is what a solve would have stored; use local with identity orientation for simplicity
        body[0:4] = (0, 0, 0, 1)
        body[24:31] = body[16:23]
    lane = np.asarray(small_scenes.prestep_for(rng, type_id, pa, pb), np.float32)
    if name in ("BallSocket", "SwivelHinge", "Hinge", "BallSocketServo", "PointOnLineServo"):
        # these act at the joint anchor: offsets are what they are; momentum about the origin is conserved only if both impulses act at one
        # world point, which holds when anchorA == anchorB. Choose LocalOffsetB so that the anchors coincide.
        off_a = lane[0:3]
        off_b_index = 3 if name in ("BallSocket", "BallSocketServo", "PointOnLineServo") else 6  # (PointOnLineServo: zero error puts A's lever arm on B's anchor)
        lane[off_b_index:off_b_index + 3] = (pa + off_a) - pb
    if name == "Weld":  # its linear rows act at B's centre with lever arm LocalOffset on A (Weld.cs:87-112): one world point iff LocalOffset = pB - pA
        lane[0:3] = pb - pa
    acc = rng.uniform(0, 0.02, TYPE_TABLE[type_id][2]).astype(np.float32)
    bodies0 = np.stack([a, b])
    p0, l0 = _momentum(bodies0)
    oracle_ffi.constraint_iterate(type_id, a, b, lane, acc, 1 / 60, 3)
    p1, l1 = _momentum(np.stack([a, b]))
    assert np.allclose(p0, p1, atol=2e-5), (name, p0, p1)
    if name != "AngularAxisGearMotor":  # a gear pair: the torques differ by the velocity scale (AngularAxisGearMotor.cs:92-96), the frame would carry the rest
        assert np.allclose(l0, l1, atol=5e-5), (name, l0, l1)


def test_ball_socket_removes_anchor_velocity():
    rng = np.random.default_rng(3)
    a, b = small_scenes.random_dynamic_body(rng, (0, 0, 0)), small_scenes.random_dynamic_body(rng, (1, 0, 0))
    for body in (a, b):
        body[0:4] = (0, 0, 0, 1)
        body[24:31] = body[16:23]
    lane = np.asarray([0.5, 0, 0, -0.5, 0, 0, 2 * np.pi * 120, 2.0], np.float32)  # stiff, anchors coincide => zero position error
    acc = np.zeros(3, np.float32)
    oracle_ffi.constraint_iterate(22, a, b, lane, acc, 1 / 60, 20)
    va = a[8:11] + np.cross(a[12:15], [0.5, 0, 0])
    vb = b[8:11] + np.cross(b[12:15], [-0.5, 0, 0])
    assert np.abs(va - vb).max() < 1e-3


def test_contact_stops_approach_and_never_pulls():
    a = make_body(position=(0, 0.5, 0), linear=(0.3, -2.0, 0.1), inverse_inertia=(6, 0, 6, 0, 0, 6))
    a[24:31] = a[16:23]
    lane = []
    for dx, dz in ((-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5)):
        lane += [dx, -0.5, dz, 0.0]
    lane += [0, 1, 0, 1.0, 2 * np.pi * 30, 2.0, 2.0]
    lane = np.asarray(lane, np.float32)
    acc = np.zeros(7, np.float32)
    oracle_ffi.constraint_iterate(3, a, None, lane, acc, 1 / 60, 10)
    assert a[9] > -0.05  # approach velocity along the normal removed (soft constraint: small residual)
    assert np.all(acc[2:6] >= 0)  # penetration impulses are clamped non-negative (PenetrationLimit.cs:23)
    assert np.hypot(acc[0], acc[1]) <= 1.0 * 0.25 * acc[2:6].sum() * 4 + 1e-6  # friction cone


def test_threads_and_fast_build_are_bitwise_identical():
    sc = small_scenes.random_graph_scene(9, 400, 1500, sorted(TYPE_TABLE.keys()))
    sd, cb = SolveDescription(2, 3), PoseIntegratorCallbacks()
    ref = sc.copy()
    oracle_ffi.solve(ref, 1 / 60, sd, cb)
    for threads, fast in ((4, False), (1, True), (3, True)):
        other = sc.copy()
        oracle_ffi.solve(other, 1 / 60, sd, cb, threads=threads, fast=fast)
        assert np.array_equal(ref.bodies.view(np.int32), other.bodies.view(np.int32)), (threads, fast)


def test_unconstrained_and_kinematic_integration_modes():
    """IntegrateBundlesAfterSubstepping semantics (PoseIntegrator.cs:537-693): unconstrained bodies take one step of dt (velocity -> pose)
    unless AllowSubstepsForUnconstrainedBodies; kinematics keep their velocity unless IntegrateVelocityForKinematics."""
    from bepuphysics2_amd.scene import SceneBuilder
    sb = SceneBuilder()
    sb.add_body(make_body(position=(0, 0, 0), linear=(1, 0, 0)))
    sb.add_body(make_body(position=(5, 0, 0), linear=(0, 1, 0), inverse_inertia=(0,) * 6, inverse_mass=0))
    dt = np.float32(1 / 60)
    for allow in (False, True):
        sc = sb.build()
        cb = PoseIntegratorCallbacks(gravity=(0, -10, 0), linear_damping=0, angular_damping=0, allow_substeps_for_unconstrained_bodies=allow)
        oracle_ffi.solve(sc, float(dt), SolveDescription(1, 4), cb)
        if not allow:
            vy = np.float32(0) + np.float32(-10) * dt
            assert sc.bodies[0, 9] == vy and sc.bodies[0, 5] == np.float32(0) + vy * dt
        else:
            assert abs(sc.bodies[0, 9] - (-10 / 60)) < 1e-6
        assert sc.bodies[1, 9] == 1 and abs(sc.bodies[1, 5] - 1 / 60) < 1e-7  # kinematic: velocity untouched, pose advanced


def _angular_momentum(body):
    """World angular momentum of a BodyDynamics record: R * I_local * R^T * w, I_local = inverse of the stored local inverse inertia."""
    x, y, z, w = [float(v) for v in body[0:4]]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    xx, yx, yy, zx, zy, zz = [float(v) for v in body[16:22]]
    inv_local = np.array([[xx, yx, zx], [yx, yy, zy], [zx, zy, zz]])
    return R @ np.linalg.inv(inv_local) @ R.T @ body[12:15].astype(np.float64)


@pytest.mark.parametrize("mode", [1, 2])
def test_conserving_angular_modes_keep_the_momentum_of_a_free_body(mode):
    """PoseIntegrator.cs:193-253: an unconstrained asymmetric body, no gravity, no damping. ConserveMomentum keeps R I R^T w to rounding every
    step; the gyroscopic mode is an implicit approximation (one Newton step) and drifts slowly; Nonconserving drifts visibly."""
    from bepuphysics2_amd.scene import SceneBuilder, make_body, PoseIntegratorCallbacks, SolveDescription
    drift = {}
    for m in (0, mode):
        sb = SceneBuilder()
        sb.add_body(make_body(position=(0, 0, 0), orientation=(0.1, 0.2, 0.3, 0.927), angular=(1.0, 2.0, 0.5), inverse_inertia=(1.0, 0.05, 0.25, 0.02, 0.03, 4.0), inverse_mass=1.0))
        scene = sb.build()
        q = scene.bodies[0, 0:4]
        scene.bodies[0, 0:4] = q / np.linalg.norm(q)
        cb = PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0.0, angular_damping=0.0, angular_integration_mode=m)
        L0 = _angular_momentum(scene.bodies[0])
        for _ in range(60):
            oracle_ffi.solve(scene, 1 / 60, SolveDescription(1, 1), cb)
        drift[m] = float(np.linalg.norm(_angular_momentum(scene.bodies[0]) - L0) / np.linalg.norm(L0))
    assert drift[0] > 0.05                       # plain integration does not conserve it
    if mode == 1:
        assert drift[mode] < 1e-4, drift         # exact up to rounding: w' = I_world'^-1 (R I R^T w)
    else:
        assert drift[mode] < drift[0] / 2, drift  # one implicit Newton step per frame: damped ("applies a damping effect", PoseIntegrator.cs:34), still far better


def test_gyroscopic_step_matches_an_independent_transcription():
    """One unconstrained step in ConserveMomentumWithGyroscopicTorque mode against a float64 numpy transcription of PoseIntegrator.cs:209-253 written
    from the C# separately from oracle/bepu_math.h (row-vector convention, Matrix3x3Wide.CreateFromQuaternion rows)."""
    from bepuphysics2_amd.scene import SceneBuilder, make_body, PoseIntegratorCallbacks, SolveDescription

    def rows(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)],
                         [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)],
                         [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]])

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    sb = SceneBuilder()
    sb.add_body(make_body(orientation=(0.1, 0.2, 0.3, 0.927), angular=(1.0, 2.0, 0.5), inverse_inertia=(1.0, 0.05, 0.25, 0.02, 0.03, 4.0)))
    scene = sb.build()
    scene.bodies[0, 0:4] /= np.linalg.norm(scene.bodies[0, 0:4])
    b0 = scene.bodies[0].astype(np.float64).copy()
    oracle_ffi.solve(scene, 1 / 60, SolveDescription(1, 1), PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0, angular_damping=0, angular_integration_mode=2))
    dt = float(np.float32(1 / 60))
    q0, w = b0[0:4], b0[12:15]
    xx, yx, yy, zx, zy, zz = b0[16:22]
    inertia = np.linalg.inv(np.array([[xx, yx, zx], [yx, yy, zy], [zx, zy, zz]]))
    speed = np.linalg.norm(w)
    dq = np.array([*(w * (np.sin(speed * dt * 0.5) / speed)), np.cos(speed * dt * 0.5)])
    a, b = q0, dq  # QuaternionWide.ConcatenateWithoutOverlap(a, b): a then b
    q1 = np.array([a[3] * b[0] + a[0] * b[3] + a[2] * b[1] - a[1] * b[2], a[3] * b[1] + a[1] * b[3] + a[0] * b[2] - a[2] * b[0],
                   a[3] * b[2] + a[2] * b[3] + a[1] * b[0] - a[0] * b[1], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])
    q1 /= np.linalg.norm(q1)
    m = rows(q1)
    lw = m @ w                      # TransformByTransposed
    momentum = lw @ inertia
    jac = inertia + dt * (skew(lw) @ inertia - skew(momentum))
    lw1 = lw - (dt * np.cross(momentum, lw)) @ np.linalg.inv(jac)
    assert np.allclose(scene.bodies[0, 12:15], lw1 @ m, rtol=2e-6, atol=1e-6)
    assert np.allclose(scene.bodies[0, 0:4], q1, rtol=0, atol=1e-6)


def test_conserving_modes_leave_kinematic_and_locked_bodies_alone():
    """FallbackIfInertiaIncompatible (PoseIntegrator.cs:179-190): inverting a zero inverse inertia gives inf/NaN, the previous velocity is kept."""
    from bepuphysics2_amd.scene import SceneBuilder, make_body, PoseIntegratorCallbacks, SolveDescription
    for mode in (1, 2):
        sb = SceneBuilder()
        sb.add_body(make_body(angular=(0.3, -0.2, 0.9), inverse_inertia=(0,) * 6, inverse_mass=0.0))
        scene = sb.build()
        oracle_ffi.solve(scene, 1 / 60, SolveDescription(1, 2), PoseIntegratorCallbacks(angular_integration_mode=mode))
        assert np.array_equal(scene.bodies[0, 12:15], np.asarray([0.3, -0.2, 0.9], dtype=np.float32))
        assert np.isfinite(scene.bodies).all()


def test_weld_removes_relative_motion():
    """Weld.cs:123-204: with zero position / orientation error the six rows drive wA - wB and vA + wA x offset - vB to zero (rigid coupling)."""
    rng = np.random.default_rng(77)
    pa, pb = np.asarray([0.2, -0.1, 0.3], np.float32), np.asarray([0.9, 0.4, -0.2], np.float32)
    a, b = small_scenes.random_dynamic_body(rng, pa), small_scenes.random_dynamic_body(rng, pb)
    for body in (a, b):
        body[0:4] = (0, 0, 0, 1)
        body[24:31] = body[16:23]
    lane = np.asarray(list(pb - pa) + [0, 0, 0, 1] + small_scenes.spring(30.0, 1.0), np.float32)
    acc = np.zeros(6, np.float32)
    before = np.abs(a[12:15] - b[12:15]).max()
    oracle_ffi.constraint_iterate(31, a, b, lane, acc, 1 / 60, 40)
    assert before > 0.1  # the pair started with plenty of relative motion
    rel_ang = a[12:15] - b[12:15]
    rel_lin = a[8:11] + np.cross(a[12:15], pb - pa) - b[8:11]
    assert np.abs(rel_ang).max() < 2e-3 and np.abs(rel_lin).max() < 2e-3, (rel_ang, rel_lin)


def _identity_pair(rng, pa, pb):
    a, b = small_scenes.random_dynamic_body(rng, pa), small_scenes.random_dynamic_body(rng, pb)
    for body in (a, b):
        body[0:4] = (0, 0, 0, 1)
        body[24:31] = body[16:23]
    return a, b


def test_widened_motors_and_servos_reach_their_targets():
    """Behavioural pins of the SURVEY 8(f) types (the oracle/device parity tests cannot catch a transcription error shared by both sides)."""
    rng = np.random.default_rng(91)
    strong_motor = [FLOAT_MAX, 1e6]  # MotorSettings{MaximumForce, Damping}: practically rigid
    pa, pb = np.asarray([0, 0, 0], np.float32), np.asarray([1, 0.5, 0], np.float32)
    zero = np.zeros(32, np.float32)
    # OneBodyAngularMotor (43): the body's angular velocity becomes the target velocity
    a, _ = _identity_pair(rng, pa, pb)
    lane = np.asarray([0.3, -0.7, 0.2] + strong_motor, np.float32)
    oracle_ffi.constraint_iterate(43, a, zero.copy(), lane, np.zeros(3, np.float32), 1 / 60, 30)
    assert np.allclose(a[12:15], [0.3, -0.7, 0.2], atol=1e-3)
    # OneBodyLinearMotor (45): the velocity of the grabbed point becomes the target velocity
    a, _ = _identity_pair(rng, pa, pb)
    off = np.asarray([0.2, 0.1, -0.3], np.float32)
    lane = np.asarray(list(off) + [0.5, 0.1, -0.4] + strong_motor, np.float32)
    oracle_ffi.constraint_iterate(45, a, zero.copy(), lane, np.zeros(3, np.float32), 1 / 60, 30)
    assert np.allclose(a[8:11] + np.cross(a[12:15], off), [0.5, 0.1, -0.4], atol=1e-3)
    # TwistMotor (28) / AngularAxisMotor (41): relative angular velocity about the axis becomes the target
    for type_id, lane in ((28, [0, 0, 1, 0, 0, 1, 0.8] + strong_motor), (41, [0, 0, 1, 0.8] + strong_motor)):
        a, b = _identity_pair(rng, pa, pb)
        oracle_ffi.constraint_iterate(type_id, a, b, np.asarray(lane, np.float32), np.zeros(1, np.float32), 1 / 60, 40)
        rel = float(a[14] - b[14])
        assert abs(abs(rel) - 0.8) < 2e-3, (type_id, rel)  # sign conventions differ between the two (TwistMotor.cs:93 vs AngularAxisMotor.cs:91)
    # BallSocketMotor (52): anchor velocity difference equals -R_A * target (BallSocketMotor.cs:80-81)
    a, b = _identity_pair(rng, pa, pb)
    lane = np.asarray([0.1, 0.2, -0.1, 0.3, 0.0, -0.2] + strong_motor, np.float32)
    oracle_ffi.constraint_iterate(52, a, b, lane, np.zeros(3, np.float32), 1 / 60, 40)
    off_b = lane[0:3]
    off_a = (pb - pa) + off_b
    rel = (a[8:11] + np.cross(a[12:15], off_a)) - (b[8:11] + np.cross(b[12:15], off_b))
    assert np.allclose(rel, -lane[3:6], atol=2e-3), rel
    # DistanceLimit (34): inside [min, max] with no approach velocity the inequality stays inactive
    a, b = _identity_pair(rng, pa, pb)
    for body in (a, b):
        body[8:15] = 0
    before = np.concatenate([a[8:15], b[8:15]]).copy()
    dist = float(np.linalg.norm(pb - pa))
    lane = np.asarray([0, 0, 0, 0, 0, 0, dist - 0.5, dist + 0.5] + small_scenes.spring(30.0, 1.0), np.float32)
    acc = np.zeros(1, np.float32)
    oracle_ffi.constraint_iterate(34, a, b, lane, acc, 1 / 60, 5)
    assert acc[0] == 0 and np.array_equal(np.concatenate([a[8:15], b[8:15]]), before)
    # DistanceServo (33): the anchors' separation speed follows the (clamped) bias velocity sign: too far apart => they approach
    a, b = _identity_pair(rng, pa, pb)
    for body in (a, b):
        body[8:15] = 0
    lane = np.asarray([0, 0, 0, 0, 0, 0, dist * 0.5, FLOAT_MAX, 0.0, FLOAT_MAX] + small_scenes.spring(30.0, 1.0), np.float32)
    oracle_ffi.constraint_iterate(33, a, b, lane, np.zeros(1, np.float32), 1 / 60, 10)
    direction = (pb - pa) / dist
    assert float(np.dot(b[8:11] - a[8:11], direction)) < -0.1


def test_linear_axis_and_line_constraints_reach_their_targets():
    """Behavioural pins of LinearAxisServo/Motor/Limit, PointOnLineServo and AngularAxisGearMotor (identity orientations, anchors at the centres)."""
    rng = np.random.default_rng(93)
    strong_motor = [FLOAT_MAX, 1e6]
    free_servo = [FLOAT_MAX, 0.0, FLOAT_MAX]
    stiff = small_scenes.spring(30.0, 1.0)
    pa, pb = np.asarray([0, 0, 0], np.float32), np.asarray([0.3, 1.0, -0.2], np.float32)
    normal = [0.0, 1.0, 0.0]
    # LinearAxisMotor (39): csi drives dot(vA - vB, n) + ... to -TargetVelocity (LinearAxisMotor.cs:99-101): B separates from A's plane at TargetVelocity
    a, b = _identity_pair(rng, pa, pb)
    lane = np.asarray([0, 0, 0, 0, 0, 0] + normal + [0.6] + strong_motor, np.float32)
    oracle_ffi.constraint_iterate(39, a, b, lane, np.zeros(1, np.float32), 1 / 60, 40)
    anchor_b_from_a = pb - pa
    rel = (b[8:11] - (a[8:11] + np.cross(a[12:15], anchor_b_from_a)))
    assert abs(float(rel[1]) - 0.6) < 2e-3, rel
    # LinearAxisServo (38): plane offset 1.0 with target 0.4 => B approaches the plane along the normal
    a, b = _identity_pair(rng, pa, pb)
    for body in (a, b):
        body[8:15] = 0
    lane = np.asarray([0, 0, 0, 0, 0, 0] + normal + [0.4] + free_servo + stiff, np.float32)
    oracle_ffi.constraint_iterate(38, a, b, lane, np.zeros(1, np.float32), 1 / 60, 10)
    assert float(b[9] - a[9]) < -0.1
    # LinearAxisLimit (40): inside [min, max] and at rest nothing happens; beyond max and at rest the pair is pulled back
    a, b = _identity_pair(rng, pa, pb)
    for body in (a, b):
        body[8:15] = 0
    before = np.concatenate([a[8:15], b[8:15]]).copy()
    acc = np.zeros(1, np.float32)
    oracle_ffi.constraint_iterate(40, a, b, np.asarray([0, 0, 0, 0, 0, 0] + normal + [0.5, 1.5] + stiff, np.float32), acc, 1 / 60, 5)
    assert acc[0] == 0 and np.array_equal(np.concatenate([a[8:15], b[8:15]]), before)
    oracle_ffi.constraint_iterate(40, a, b, np.asarray([0, 0, 0, 0, 0, 0] + normal + [0.0, 0.5] + stiff, np.float32), acc, 1 / 60, 5)
    assert acc[0] > 0 and float(b[9] - a[9]) < -0.05
    # PointOnLineServo (37): B's anchor stays on a line through A along x: the relative anchor velocity loses its y/z components, x is free
    a, b = _identity_pair(rng, pa, pb)
    lane = np.asarray([0, 0, 0] + list(pa - pb) + [1.0, 0.0, 0.0] + free_servo + stiff, np.float32)  # LocalOffsetB puts B's anchor on A's centre: zero error
    before_x = float((b[8:11] + np.cross(b[12:15], pa - pb) - a[8:11])[0])
    oracle_ffi.constraint_iterate(37, a, b, lane, np.zeros(2, np.float32), 1 / 60, 40)
    rel = b[8:11] + np.cross(b[12:15], pa - pb) - a[8:11]
    assert np.abs(rel[1:3]).max() < 2e-3, rel
    assert abs(before_x) > 0.05 and abs(float(rel[0])) > 0.01  # sliding along the line is left alone
    # AngularAxisGearMotor (54): from a zero accumulated impulse one Solve makes dot(wA, axis) * scale == dot(wB, axis) (:92-108)
    a, b = _identity_pair(rng, pa, pb)
    lane = np.asarray([0, 0, 1, 2.5] + strong_motor, np.float32)
    oracle_ffi.constraint_iterate(54, a, b, lane, np.zeros(1, np.float32), 1 / 60, 1)
    assert abs(float(a[14]) * 2.5 - float(b[14])) < 2e-3, (a[14], b[14])


def test_nonconvex_contacts_stop_approach_per_contact_normal():
    """ContactNonconvexCommon.cs:208-228: each contact has its own normal; penetration impulses stay nonnegative, friction stays inside its cone."""
    a = make_body(position=(0, 0.5, 0), linear=(0.3, -2.0, 0.1), inverse_inertia=(6, 0, 6, 0, 0, 6))
    a[24:31] = a[16:23]
    lane = [1.0] + small_scenes.spring(30.0, 1.0) + [2.0]
    normals = [(0, 1, 0), (0.1, 0.99, 0.0), (0.0, 0.99, 0.1), (-0.1, 0.99, 0.0)]
    for (dx, dz), n in zip(((-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5)), normals):
        n = np.asarray(n, np.float64) / np.linalg.norm(n)
        lane += [dx, -0.5, dz, 0.0] + list(n)
    acc = np.zeros(12, np.float32)
    zero = np.zeros(32, np.float32)
    oracle_ffi.constraint_iterate(10, a, zero, np.asarray(lane, np.float32), acc, 1 / 60, 30)
    assert a[9] > -0.05  # the approach along the (mostly +y) normals is stopped
    for c in range(4):
        assert acc[3 * c + 2] >= 0
        assert np.hypot(acc[3 * c], acc[3 * c + 1]) <= 1.0 * acc[3 * c + 2] * (1 + 1e-5) + 1e-7


def _solve_free_bodies(positions, masses, type_id, prestep, frames=40, substeps=4):
    """A handful of point-like dynamic bodies, one constraint over all of them, no gravity or damping: returns the scene after `frames` solves."""
    from bepuphysics2_amd.scene import SceneBuilder
    sb = SceneBuilder()
    rng = np.random.default_rng(5)
    hs = []
    for pos, m in zip(positions, masses):
        hs.append(sb.add_body(make_body(position=pos, linear=rng.uniform(-0.3, 0.3, 3), inverse_mass=1.0 / m)))
    sb.add_constraint(type_id, hs, prestep)
    scene = sb.build()
    cb = PoseIntegratorCallbacks(gravity=(0, 0, 0), linear_damping=0.0, angular_damping=0.0)
    start = scene.copy()
    for _ in range(frames):
        oracle_ffi.solve(scene, 1 / 60, SolveDescription(1, substeps), cb)
    return start, scene


def test_center_distance_area_and_volume_constraints_hold_their_targets():
    """The four types restated on the portable branch of MathHelper.FastReciprocal*: the constrained quantity converges to its target and the impulses
    (along the centre line; jacobians summing to zero) leave the total linear momentum untouched."""
    stiff = small_scenes.spring(30.0, 1.0)
    masses = [1.0, 2.0, 0.5, 1.5]

    def momentum(sc, n):
        return sum(m * sc.bodies[i, 8:11].astype(np.float64) for i, m in zip(range(n), masses))

    # CenterDistanceConstraint (35): |pB - pA| -> TargetDistance
    start, end = _solve_free_bodies([(0, 0, 0), (1.5, 0.2, -0.1)], masses[:2], 35, [2.0] + stiff)
    assert abs(float(np.linalg.norm(end.bodies[1, 4:7] - end.bodies[0, 4:7])) - 2.0) < 2e-2
    assert np.allclose(momentum(start, 2), momentum(end, 2), atol=1e-4)
    # CenterDistanceLimit (55): outside [min, max] the pair is brought back to the nearer bound, inside nothing happens
    start, end = _solve_free_bodies([(0, 0, 0), (3.0, 0, 0)], masses[:2], 55, [0.5, 2.0] + stiff)
    assert float(np.linalg.norm(end.bodies[1, 4:7] - end.bodies[0, 4:7])) < 2.3
    # AreaConstraint (36): |ab x ac| -> TargetScaledArea
    tri = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    start, end = _solve_free_bodies(tri, masses[:3], 36, [2.0] + stiff)
    a, b, c = (end.bodies[i, 4:7].astype(np.float64) for i in range(3))
    assert abs(float(np.linalg.norm(np.cross(b - a, c - a))) - 2.0) < 5e-2
    assert np.allclose(momentum(start, 3), momentum(end, 3), atol=1e-4)
    # VolumeConstraint (32): dot(ab x ac, ad) -> TargetScaledVolume
    tet = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    start, end = _solve_free_bodies(tet, masses, 32, [2.5] + stiff)
    a, b, c, d = (end.bodies[i, 4:7].astype(np.float64) for i in range(4))
    assert abs(float(np.dot(np.cross(b - a, c - a), d - a)) - 2.5) < 8e-2
    assert np.allclose(momentum(start, 4), momentum(end, 4), atol=1e-4)


FLOAT_MAX = float(np.finfo(np.float32).max)
