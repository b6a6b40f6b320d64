"""bepuhip_set_velocity_model on the device against the oracle (which tests/test_velocity_models.py checks against its independent second restatement): the per-body
gravity of Demos/Demos/PerBodyGravityDemo.cs and the radial gravity of Demos/Demos/PlanetDemo.cs in every stage that calls IntegrateVelocity — substep integration of
constrained bodies on all three schedules, the kinematic prepass, IntegrateAfterSubstepping for unconstrained bodies, PredictBoundingBoxes — bit for bit."""
import numpy as np
import pytest

import oracle_ffi
import parity_util as pu
import small_scenes
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription
from test_velocity_models import model_callbacks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("use_clusters", [True, False])
def test_velocity_models_on_island_scenes(hip_solver_factory, model, use_clusters):
    rng = np.random.default_rng(17 + model)
    scene = small_scenes.island_scene(11, islands=60, bodies_per_island=9, constraints_per_island=22, type_ids=[4, 7, 22, 23, 25, 47, 0, 3, 31, 35])
    for sd, kw in ((SolveDescription(2, 3), {}),
                   (SolveDescription(1, 4, velocity_iteration_scheduler=lambda s: [1, 3, 2, 1][s]), {"integrate_velocity_for_kinematics": True, "allow_substeps_for_unconstrained_bodies": True}),
                   (SolveDescription(1, 2), {"angular_integration_mode": 1}), (SolveDescription(2, 2), {"angular_integration_mode": 2, "integrate_velocity_for_kinematics": True})):
        cb = model_callbacks(model, scene, rng, **kw)
        ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=3, threads=4)
        solver = hip_solver_factory(use_clusters=use_clusters)
        try:
            got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=3)
        except Exception as e:
            raise AssertionError(f"model {model}, clusters {use_clusters}, {kw}: {e}") from e
        assert (solver.schedule() == 1) == use_clusters
        m = pu.compare_scenes(ref, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (model, use_clusters, kw, m)


@pytest.mark.parametrize("model", [1, 2])
def test_velocity_models_on_a_split_plan(hip_solver_factory, monkeypatch, model):
    """One island cut into clusters: the home cluster integrates a shared body (and publishes the model's velocity in its record), ghost copies only follow its pose."""
    monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "14")
    rng = np.random.default_rng(5)
    scene = small_scenes.random_graph_scene(21, 3000, 7000, [4, 5, 6, 7, 22, 25, 47, 0, 3], kinematic_fraction=0.05)
    sd = SolveDescription(1, 3, velocity_iteration_scheduler=lambda s: [2, 1, 2][s])
    cb = model_callbacks(model, scene, rng, integrate_velocity_for_kinematics=True)
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=2, threads=4)
    solver = hip_solver_factory()
    got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=2)
    assert solver.schedule() == 2 and solver.cluster_cycles().size > 1
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (model, m)


def test_the_model_follows_the_context_and_a_stale_table_is_refused(hip_solver_factory):
    from bepuphysics2_amd.native import BepuHipError
    scene = small_scenes.island_scene(3, islands=10, bodies_per_island=6, constraints_per_island=10, type_ids=[4, 22])
    sd = SolveDescription(1, 2)
    solver = hip_solver_factory()
    solver.upload(scene, sd.fallback_batch_threshold)
    with pytest.raises(BepuHipError):  # one value per body, or the solve is refused before anything runs
        solver.solve(1 / 60, sd, PoseIntegratorCallbacks(velocity_model=1, body_gravity=np.zeros(scene.body_count - 1, np.float32)))
    # back to the uniform model: the same context solves as a fresh one does
    cb = PoseIntegratorCallbacks()
    solver.solve(1 / 60, sd, PoseIntegratorCallbacks(velocity_model=2, planet_gravity=100.0))
    solver.upload(scene, sd.fallback_batch_threshold)
    solver.solve(1 / 60, sd, cb)
    got = scene.copy()
    solver.download(got)
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=1)
    assert pu.compare_scenes(ref, got)["bodies_bit_exact"]


@pytest.mark.parametrize("model", [1, 2])
def test_predict_bounding_boxes_evaluates_the_model(hip_solver_factory, model):
    from test_bounds import _random_bodies, _random_collidables  # the stage's own test scene
    rng = np.random.default_rng(9)
    bodies, collidables = _random_bodies(rng, 600), _random_collidables(rng, 600)
    cb = PoseIntegratorCallbacks(velocity_model=1, body_gravity=rng.uniform(-30, 30, bodies.shape[0]).astype(np.float32)) if model == 1 else \
        PoseIntegratorCallbacks(velocity_model=2, planet_center=(1.0, 2.0, -1.0), planet_gravity=900.0, integrate_velocity_for_kinematics=True)
    ref = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, collidables)
    solver = hip_solver_factory()
    solver.set_bodies(bodies)
    got = solver.predict_bounding_boxes(1 / 60, cb, collidables)
    assert np.array_equal(ref.view(np.uint8), got.view(np.uint8))


@pytest.mark.parametrize("schedule", ["islands", "split", "batches"])
def test_substep_events_are_raised_around_every_substep_and_may_rewrite_state(hip_solver_factory, monkeypatch, schedule):
    """Solver.SubstepStarted / SubstepEnded (Solver.cs:125-146, raised at Solver_Solve.cs:1425 / :1478) through bepuhip_solve_with_substep_events: handlers that change
    nothing leave bepuhip_solve's bits; a SubstepStarted handler that moves a kinematic body's velocity every substep (what the reference's users do there) gives
    what the oracle gives when the same writes are made between single-substep solves... which the reference cannot express — so the second half is checked
    against the device's own plain solve of a scene whose kinematic body carries the velocity the handler writes first."""
    # Round 5: a context on an island plan raises the events between launches of the island kernel, one substep per launch (whole-island and split plans);
    # BEPUHIP_EVENT_CLUSTERS=0 keeps round 4's launch-per-batch form covered.
    if schedule == "batches":
        monkeypatch.setenv("BEPUHIP_EVENT_CLUSTERS", "0")
    if schedule == "split":
        monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "8")
        monkeypatch.setenv("BEPUHIP_FORCE_SPLIT", "64")
        scene = small_scenes.random_graph_scene(17, 1200, 3000, [4, 5, 7, 22, 25, 47, 0], kinematic_fraction=0.05)
    else:
        scene = small_scenes.island_scene(8, islands=30, bodies_per_island=8, constraints_per_island=18, type_ids=[4, 7, 22, 25, 47, 0])
    sd = SolveDescription(1, 4, velocity_iteration_scheduler=lambda s: [2, 1, 1, 2][s])
    cb = PoseIntegratorCallbacks(integrate_velocity_for_kinematics=schedule == "split")
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=2, threads=4)
    solver = hip_solver_factory()
    solver.upload(scene, sd.fallback_batch_threshold)
    assert solver.schedule() == {"islands": 1, "split": 2, "batches": 1}[schedule]
    seen = []
    for _ in range(2):
        solver.solve_with_substep_events(1 / 60, sd, cb, started=lambda s: seen.append(("started", s)), ended=lambda s: seen.append(("ended", s)))
    got = scene.copy()
    solver.download(got)
    assert seen == [(kind, s) for _ in range(2) for s in range(4) for kind in ("started", "ended")]
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
    # a handler that writes state: every body's linear velocity is zeroed when substep 2 starts — the device must see the write (its results differ from the plain solve's)
    solver.upload(scene, sd.fallback_batch_threshold)

    def freeze(substep):
        if substep == 2:
            bodies = solver.get_bodies(scene.body_count)
            bodies[:, 8:11] = 0.0
            solver.update_bodies(0, bodies)

    solver.solve_with_substep_events(1 / 60, sd, cb, started=freeze)
    frozen = solver.get_bodies(scene.body_count)
    plain = pu.run_hip(hip_solver_factory(), scene, 1 / 60, sd, cb, frames=1).bodies
    assert not np.array_equal(frozen[:, 8:11], plain[:, 8:11]) and np.isfinite(frozen).all()
    with pytest.raises(Exception):  # structural changes inside a handler are refused
        solver.solve_with_substep_events(1 / 60, sd, cb, started=lambda s: solver.remove_constraint(0, scene.batches[0][0].type_id, 0))
