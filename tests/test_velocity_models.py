"""IPoseIntegratorCallbacks.IntegrateVelocity beyond the demo callbacks' gravity + damping (VERDICT r3 missing #4): the per-body gravity of
Demos/Demos/PerBodyGravityDemo.cs:57-88 and the radial gravity of Demos/Demos/PlanetDemo.cs:36-47, restated twice (oracle/ scalar per lane; oracle/wide in the C#'s
own shape, bodyIndices / position / dt arguments and all) — the two must agree bit for bit in every stage that calls the callback."""
import numpy as np
import pytest

import oracle_ffi
import parity_util as pu
import small_scenes
import wide_ffi
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription


def model_callbacks(model, scene, rng, **kw):
    if model == 1:
        return PoseIntegratorCallbacks(velocity_model=1, body_gravity=rng.uniform(-20, 5, scene.body_count).astype(np.float32), **kw)
    return PoseIntegratorCallbacks(velocity_model=2, planet_center=(0.5, -1.0, 0.25), planet_gravity=350.0, **kw)


@pytest.mark.parametrize("model", [1, 2])
def test_both_oracles_agree_on_every_stage_that_calls_the_callback(model):
    rng = np.random.default_rng(3 + model)
    scene = small_scenes.island_scene(11, islands=40, bodies_per_island=9, constraints_per_island=22, type_ids=[4, 7, 22, 23, 25, 47, 0, 3, 31, 35])
    for sd, kw in ((SolveDescription(2, 3), {}),
                   (SolveDescription(1, 4, velocity_iteration_scheduler=lambda s: [1, 3, 2, 1][s]), {"integrate_velocity_for_kinematics": True, "allow_substeps_for_unconstrained_bodies": True}),
                   (SolveDescription(1, 2), {"angular_integration_mode": 1}), (SolveDescription(2, 2), {"angular_integration_mode": 2, "integrate_velocity_for_kinematics": True})):
        cb = model_callbacks(model, scene, rng, **kw)
        a, b = scene.copy(), scene.copy()
        for _ in range(3):
            oracle_ffi.solve(a, 1 / 60, sd, cb, threads=3)
            wide_ffi.solve(b, 1 / 60, sd, cb, threads=2)
        m = pu.compare_scenes(a, b)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (model, kw, m)
        assert np.isfinite(a.bodies[:, :15]).all()
        assert float(np.abs(a.bodies[:, 8:11] - scene.bodies[:, 8:11]).max()) > 1e-3  # the model did something


def test_a_constant_per_body_table_is_uniform_gravity_without_damping():
    """linear.Y += g * dt against (linear + (0, g, 0) * dt) * pow(1 - 0, dt): the same bits, body for body — pins the table's indexing and the dt the callback gets
    (substep dt inside the solve, the final pass's own dt for unconstrained bodies)."""
    scene = small_scenes.island_scene(5, islands=25, bodies_per_island=8, constraints_per_island=18, type_ids=[4, 6, 22, 25, 30, 47])
    sd = SolveDescription(2, 3)
    uniform = PoseIntegratorCallbacks(gravity=(0.0, -7.5, 0.0), linear_damping=0.0, angular_damping=0.0, allow_substeps_for_unconstrained_bodies=True)
    table = PoseIntegratorCallbacks(velocity_model=1, body_gravity=np.full(scene.body_count, -7.5, np.float32), allow_substeps_for_unconstrained_bodies=True)
    a, b = scene.copy(), scene.copy()
    for _ in range(2):
        oracle_ffi.solve(a, 1 / 60, sd, uniform)
        oracle_ffi.solve(b, 1 / 60, sd, table)
    # (+0 * dt added to X and Z by the uniform model can only turn -0 into +0)
    assert np.array_equal(a.bodies[:, :15] + 0.0, b.bodies[:, :15] + 0.0)


def test_radial_gravity_pulls_towards_the_centre_and_is_capped_inside_the_unit_sphere():
    """One unconstrained body outside the unit sphere around the centre, one inside: v -= dt * G * offset / max(1, d^3), once per frame (AllowSubstepsForUnconstrainedBodies off)."""
    from bepuphysics2_amd.scene import SceneBuilder, make_body
    sb = SceneBuilder()
    sb.add_body(make_body(position=(10, 0, 0)))
    sb.add_body(make_body(position=(0.25, 0.25, 0)))
    a = sb.add_body(make_body(position=(30, 0, 0)))
    b = sb.add_body(make_body(position=(31, 0, 0)))
    sb.add_constraint(22, [a, b], small_scenes.joint_prestep(np.random.default_rng(0), 22))  # a scene needs a constraint
    scene = sb.build()
    cb = PoseIntegratorCallbacks(velocity_model=2, planet_center=(0, 0, 0), planet_gravity=600.0)
    oracle_ffi.solve(scene, 1 / 60, SolveDescription(1, 1), cb)
    dt = np.float32(1 / 60)
    gdt = dt * np.float32(600.0)
    far = -(np.float32(10) * gdt) * (np.float32(1) / np.float32(1000))
    assert scene.bodies[0, 8] == np.float32(far) and scene.bodies[0, 9] == 0 and scene.bodies[0, 10] == 0
    near = -(np.float32(0.25) * gdt) * np.float32(1)  # d^3 < 1: the divisor is 1
    assert scene.bodies[1, 8] == np.float32(near) and scene.bodies[1, 9] == np.float32(near)
