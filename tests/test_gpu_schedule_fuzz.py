"""Deterministic slices of the device fuzzers in the GPU suite (the only failing evidence of round 3 lived in tools/ the driver never runs), and the
regression test of what that evidence turned out to be.

The fault ("Overflow waits"): work items with more predecessors than an item records waited on per-batch publish COUNTERS. Inside a fused sweep — warm start
and first velocity iteration in one claim sequence — a Solve item of batch b may publish before a WarmStart item of batch b has, the count `epoch x items of the batch` is
then reached with a warm-start item outstanding, and a cross-overflow Solve item of batch 0 started on a body whose last warm-start application had not happened
(tools/fuzz_device.py seed 81, ordinal 91: 12 % of the runs of a cold process, 0.2 % of a warm one). Fixed by waiting on the items' own pass flags. Because such a window
only opens under unusual timing, these tests run under schedule fuzzing (BEPUHIP_DEBUG_JITTER: pseudo-random naps around every item's wait and publish) — with it the
round-3 wait fails `test_overflow_items_wait_on_item_flags...` in most of its runs (tools/experiments/overflow_race/, profiles/r04_overflow_race.txt); without it, in
one run in five hundred."""
import numpy as np
import pytest

import fuzz_util as fu

pytestmark = pytest.mark.gpu


def test_overflow_items_wait_on_item_flags_inside_a_fused_sweep():
    """The scene that exposed the race: 6 whole-island clusters, three of which hold a batch-0 work item with more than kMaxPreds cross-pass predecessors."""
    p = fu.device_scene_parameters(81, 92)[91]
    scene, sd = fu.build_device_scene(p)
    import parity_util as pu
    ref = pu.run_oracle(scene, 1 / 60, sd, p["cb"], frames=p["frames"], threads=4)
    assert fu.oracle_is_finite(ref)
    wrong = []
    for run in range(48):  # every angular mode's kernel unit, natural timing and 3 x 15 jitter patterns
        q = dict(p)
        mode = run % 3
        cb = p["cb"]
        q["cb"] = type(cb)(gravity=cb.gravity, linear_damping=cb.linear_damping, angular_damping=cb.angular_damping, integrate_velocity_for_kinematics=cb.integrate_velocity_for_kinematics,
                           allow_substeps_for_unconstrained_bodies=cb.allow_substeps_for_unconstrained_bodies, angular_integration_mode=mode)
        expected = ref if mode == cb.angular_integration_mode else pu.run_oracle(scene, 1 / 60, sd, q["cb"], frames=p["frames"], threads=4)
        got, info = fu.run_device(q, scene, sd, jitter=0 if run < 3 else 1000 + run)
        assert info[0] == 1 and info[2] == 6, info  # whole-island plan, six clusters
        if not fu.exact(expected, got):
            wrong.append((run, mode))
    assert not wrong, f"runs (index, angular mode) that differ from the oracle: {wrong}"


@pytest.mark.parametrize("seed,first,count", [(81, 60, 70), (9, 0, 60), (404, 0, 60)])
def test_device_fuzz_slice(seed, first, count):
    """Scenes [first, first + count) of tools/fuzz_device.py's generator for `seed` (81: the run that found the overflow race, ordinal 91 included), odd ordinals under
    schedule fuzzing exactly as the tool runs them: every compared scene bit-identical to the oracle; nothing refused."""
    params = fu.device_scene_parameters(seed, first + count)
    compared, wrong, island, split = 0, [], 0, 0
    for ordinal in range(first, first + count):
        p = params[ordinal]
        verdict, (schedule, policy, clusters) = fu.check_device_scene(p, jitter=((seed * 7919 + ordinal) | 1) if ordinal % 2 else 0)
        island += clusters > 0
        split += bool(p["big"] and clusters > 1)
        if verdict == "diverged":
            continue
        compared += 1
        if verdict == "mismatch":
            wrong.append((ordinal, fu.describe(p)))
    assert not wrong, wrong
    assert compared >= count * 3 // 4 and island >= count // 2 and split >= 2, (compared, island, split)  # the slice really covers the island schedules


@pytest.mark.parametrize("seed", [515, 2026])
def test_split_plan_fuzz_slice_under_jitter(seed):
    """the slices above only hold the split-island scenes their generator happens to draw (two or three each). Here the twelve first BIG scenes of a
    seed's sequence — forced split plans over random subsets of all 44 type ids, conserving modes, kinematic fractions — every one under schedule fuzzing, which since
    round 5 also naps in front of the shared-record polls (acquire_shared / acquire_shared_many) and in front of every record publish (release_shared): the
    cross-workgroup hand-off protocol under timing no natural run produces. Every compared scene bit-identical to the oracle."""
    params = [p for p in fu.device_scene_parameters(seed, 400) if p["big"]][:12]
    assert len(params) == 12
    compared, wrong, split = 0, [], 0
    for ordinal, p in enumerate(params):
        p = dict(p, use_clusters=True)
        verdict, (schedule, policy, clusters) = fu.check_device_scene(p, jitter=(seed * 7919 + ordinal) | 1)
        split += schedule == 2 and clusters > 1
        if verdict == "diverged":
            continue
        compared += 1
        if verdict == "mismatch":
            wrong.append((ordinal, fu.describe(p)))
    assert not wrong, wrong
    assert split >= 10 and compared >= 8, (split, compared)


@pytest.mark.parametrize("seed,count", [(61, 24), (7, 24)])
def test_structural_fuzz_slice(seed, count):
    """`count` scenes of tools/fuzz_structural.py's generator: random add / remove / body-removal / re-plan streams, every frame equal to the oracle solving the host mirror."""
    rng = np.random.default_rng(seed)
    frames = 0
    for scene in range(count):
        stats = fu.run_structural_scene(rng, jitter=((seed * 7919 + scene) | 1) if scene % 2 else 0)
        assert stats["ok"], (scene, stats["report"])
        frames += stats["frames"]
    assert frames >= 3 * count


def test_an_addition_into_a_type_batch_that_was_empty_when_the_plan_was_built_reads_back_from_its_own_slot():
    """tools/fuzz_structural.py 6802, scene 490 (the generator's state in front of it: tests/golden/fuzz_structural_6802_490_state.json): a Contact3 type batch loses its only
    constraint, the context re-plans with the type batch empty (reserved slots only), the next frame adds a constraint to it. The host index -> device slot table of an
    EMPTY type batch used to be built with one entry per device slot instead of none, so the addition's entry landed behind `slots` zeros and index 0 pointed at device slot 0:
    the constraint was solved in its own slot and read back (and ranged-updated) from a dead one — zero prestep data, NaN impulses, bodies bit-exact as long as the contact
    stayed inactive. HostTypeBatch::perm_inverse now sizes the table by the type batch."""
    import json
    import os
    import numpy as np
    import fuzz_util as fu
    state = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_structural_6802_490_state.json")))
    rng = np.random.default_rng(0)
    rng.bit_generator.state = state
    stats = fu.run_structural_scene(rng, jitter=0)
    assert stats["ok"], stats["report"]
    assert stats["replans"] >= 1 and stats["big"]
