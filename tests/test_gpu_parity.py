"""-m gpu: parity tests proper — the HIP path through the C ABI vs the CPU oracle on identical seeded scenes.

Tolerance (BASELINE.json north_star): <= 1e-4 relative velocity error after 8 substeps. With FMA contraction off and correctly
rounded divide/sqrt on both sides we additionally *expect* bit-exact results and assert that where the reference's own
semantics make it well defined (no NaN lanes)."""
import numpy as np
import pytest

import parity_util as pu
import small_scenes
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks, SolveDescription

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # north_star: relative velocity error vs reference after 8 substeps


def _check(m):
    assert m["velocity_rel_err"] <= REL_TOL, m
    assert m["position_rel_err"] <= REL_TOL, m


def test_box_stack_bit_exact(hip_solver_factory):
    solver = hip_solver_factory()
    scene = small_scenes.box_stack_scene()
    sd, cb = SolveDescription(4, 1), PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=4)
    got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=4)
    m = pu.compare_scenes(ref, got)
    _check(m)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"], m


@pytest.mark.parametrize("type_id", sorted(TYPE_TABLE.keys()))
def test_each_type_random_graph(hip_solver_factory, type_id):
    solver = hip_solver_factory()
    scene = small_scenes.random_graph_scene(100 + type_id, 300, 700, [type_id])
    sd, cb = SolveDescription(2, 8), PoseIntegratorCallbacks()  # 8 substeps x 2 iterations
    ref = pu.run_oracle(scene, 1 / 60, sd, cb)
    got = pu.run_hip(solver, scene, 1 / 60, sd, cb)
    m = pu.compare_scenes(ref, got)
    _check(m)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def test_mixed_types_multi_frame(hip_solver_factory):
    solver = hip_solver_factory()
    scene = small_scenes.random_graph_scene(7, 2000, 6000, sorted(TYPE_TABLE.keys()))
    sd, cb = SolveDescription(2, 4), PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=3)
    got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=3)
    m = pu.compare_scenes(ref, got)
    _check(m)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


@pytest.mark.parametrize("name,a,frames", [("pyramid", 3, 2), ("pile", 8000, 2), ("ragdoll_tube", 500, 2)])
def test_reference_scene_recipes(hip_solver_factory, name, a, frames):
    """Scenes built by the C++ host mirror in the reference's insertion order (greedy batch colouring), solved with each scene's own SolveDescription."""
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene(name, a, 1 if name == "ragdoll_tube" else 0, 0, 5)
    sim.validate()
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    cb = PoseIntegratorCallbacks()
    solver = hip_solver_factory()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=frames, threads=4)
    got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=frames)
    m = pu.compare_scenes(ref, got)
    _check(m)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
