"""-m gpu: the stream schedule (the launch-per-batch sequence as one cooperative launch, csrc/bepu_stream_kernel.h) against the oracle, bit for bit."""
import numpy as np
import pytest

import parity_util as pu
import small_scenes
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

pytestmark = pytest.mark.gpu

HOT_TYPES = [0, 1, 2, 3, 4, 5, 6, 7, 22, 23, 24, 25, 26, 27, 46, 47]


def _bit_exact(ref, got):
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def _stream_solver(factory):
    return factory(use_clusters=False, use_stream=True)


@pytest.mark.parametrize("seed,bodies,constraints", [(3, 60, 150), (4, 700, 2600), (5, 3000, 9000)])
def test_stream_schedule_matches_the_oracle(hip_solver_factory, seed, bodies, constraints):
    """Random connected graphs over all sixteen hot-path types, kinematic references included: more blocks than one wavefront owns, several
    batches, several frames (so that accumulated impulses and incrementally updated depths travel between launches)."""
    scene = small_scenes.random_graph_scene(seed, bodies, constraints, HOT_TYPES)
    sd, cb = SolveDescription(3, 4), PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=3)
    _bit_exact(ref, pu.run_hip(_stream_solver(hip_solver_factory), scene, 1 / 60, sd, cb, frames=3))


def test_stream_schedule_with_kinematic_velocity_integration_and_per_substep_iterations(hip_solver_factory):
    scene = small_scenes.random_graph_scene(11, 400, 1500, HOT_TYPES, kinematic_fraction=0.15)
    sd = SolveDescription(2, 4, velocity_iteration_scheduler=lambda s: [1, 3, 2, 4][s])
    for cb in (PoseIntegratorCallbacks(integrate_velocity_for_kinematics=True), PoseIntegratorCallbacks(allow_substeps_for_unconstrained_bodies=True)):
        ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=2)
        _bit_exact(ref, pu.run_hip(_stream_solver(hip_solver_factory), scene, 1 / 60, sd, cb, frames=2))


def test_stream_schedule_equals_the_launch_per_batch_schedule_on_a_pile(hip_solver_factory):
    """BASELINE.json configs[1] at full size (100k boxes, one island) on both forms of the schedule: identical bits after 3 frames."""
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene("pile", 100000, 0, 0, 5)
    scene, sd = sim.export(), sim.solve_description()
    cb = PoseIntegratorCallbacks()
    a = pu.run_hip(hip_solver_factory(use_clusters=False), scene, 1 / 60, sd, cb, frames=3)
    b = pu.run_hip(_stream_solver(hip_solver_factory), scene, 1 / 60, sd, cb, frames=3)
    _bit_exact(a, b)


def test_scenes_outside_the_stream_schedule_fall_back(hip_solver_factory):
    """A widened type (Weld, id 28) or a conserving angular mode: the flag is accepted and the launch-per-batch schedule runs."""
    scene = small_scenes.random_graph_scene(8, 80, 200, [22, 28, 4])
    sd, cb = SolveDescription(2, 2), PoseIntegratorCallbacks()
    _bit_exact(pu.run_oracle(scene, 1 / 60, sd, cb, frames=2), pu.run_hip(_stream_solver(hip_solver_factory), scene, 1 / 60, sd, cb, frames=2))
    hot = small_scenes.random_graph_scene(9, 80, 200, [22, 23, 4])
    cb = PoseIntegratorCallbacks(angular_integration_mode=1)
    _bit_exact(pu.run_oracle(hot, 1 / 60, sd, cb, frames=2), pu.run_hip(_stream_solver(hip_solver_factory), hot, 1 / 60, sd, cb, frames=2))
