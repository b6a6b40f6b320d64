"""-m gpu: the island kernel's type-set families (round 6). A unit of the contacts family is the hot unit minus the switch cases of the eight joint types; the launcher
picks it when the scene holds nothing but convex contact manifolds (type ids 0-7: BASELINE.json configs[0] and configs[1]). Same bits by construction — asserted here on
both plan kinds, against the oracle and against the superset family (BEPUHIP_CONTACTS_FAMILY=0), together with the switch back when a joint arrives."""
import numpy as np
import pytest

import parity_util as pu
import small_scenes
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

pytestmark = pytest.mark.gpu
CONTACTS = [0, 1, 2, 3, 4, 5, 6, 7]


@pytest.fixture(autouse=True)
def plain_rows(monkeypatch):
    """The launch policy's first fifteen solves cycle through its candidates, and the non-temporal-rows candidate exists in the hot and wide families only (a contacts
    scene runs the hot unit for those launches: the next larger family, same bits): pinned to plain rows, so that every launch of these tests is the family under test."""
    monkeypatch.setenv("BEPUHIP_ROW_POLICY", "0")
    monkeypatch.setenv("BEPUHIP_SPECIALISE", "0")  # (a suite run under BEPUHIP_SPECIALISE=1 would give every context of these tests its own unit: family 3 before the tests ask for it)


def _exact(ref, got):
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


@pytest.mark.parametrize("plan", ["whole_islands", "split"])
def test_contacts_only_scene_runs_the_contacts_family_with_the_same_bits(hip_solver_factory, monkeypatch, plan):
    if plan == "split":
        monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "14")
        monkeypatch.setenv("BEPUHIP_FORCE_SPLIT", "64")
        scene = small_scenes.random_graph_scene(31, 2500, 7000, CONTACTS)
    else:
        scene = small_scenes.island_scene(32, islands=60, bodies_per_island=14, constraints_per_island=40, type_ids=CONTACTS)
    sd, cb = SolveDescription(2, 4), PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=3, threads=4)
    solver = hip_solver_factory()
    got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=3)
    assert solver.schedule() == (2 if plan == "split" else 1)
    assert solver.kernel_family() == 0, "a scene of convex contact manifolds runs the contacts family"
    _exact(ref, got)
    monkeypatch.setenv("BEPUHIP_CONTACTS_FAMILY", "0")  # the superset family on the same scene
    hot = hip_solver_factory()
    got_hot = pu.run_hip(hot, scene, 1 / 60, sd, cb, frames=3)
    assert hot.kernel_family() == 1
    _exact(ref, got_hot)
    assert np.array_equal(got.bodies.view(np.int32), got_hot.bodies.view(np.int32))


def test_a_joint_in_the_scene_keeps_the_hot_family_and_widened_types_the_wide_one(hip_solver_factory):
    sd, cb = SolveDescription(1, 3), PoseIntegratorCallbacks()
    for types, family in ((CONTACTS + [22], 1), (CONTACTS + [22, 31], 2)):
        scene = small_scenes.island_scene(33, islands=30, bodies_per_island=12, constraints_per_island=36, type_ids=types)
        solver = hip_solver_factory()
        assert solver.kernel_family() == -1
        got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=2)
        assert solver.schedule() == 1 and solver.kernel_family() == family
        _exact(pu.run_oracle(scene, 1 / 60, sd, cb, frames=2), got)


def test_the_family_follows_structural_updates(hip_solver_factory):
    """A contacts-only context that is handed a BallSocket leaves the contacts family at that launch (the family is picked per launch from the type ids present)."""
    import oracle_ffi
    from mutable_scene import MutableSolver
    rng = np.random.default_rng(5)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-4, 4, 3)) for _ in range(120)]
    ms = MutableSolver(np.stack(rows))
    for _ in range(200):
        t = CONTACTS[int(rng.integers(4, 8))]
        a, b = (int(x) for x in rng.choice(120, 2, replace=False))
        ms.add(t, [a, b], small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
    sd, cb = SolveDescription(1, 2), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    families = []
    for frame in range(4):
        if frame == 2:
            a, b = 3, 77
            lane = small_scenes.prestep_for(rng, 22, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
            bi, index, encoded = ms.add(22, [a, b], lane)
            assert solver.add_constraint(bi, 22, encoded, lane) == index
        export = ms.to_scene()
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        families.append((solver.schedule(), solver.kernel_family()))
        got = ms.to_scene()
        solver.download(got)
        _exact(export, got)
    assert families[0][1] == 0 and families[1][1] == 0
    # with the joint: the hot family if the island plan absorbed the new type batch, none (-1) if the context went to the launch-per-batch schedule, whose kernels carry every type
    assert all((s == 0 and f == -1) or (s != 0 and f == 1) for s, f in families[2:]), families


@pytest.mark.parametrize("plan", ["whole_islands", "split"])
def test_a_unit_compiled_for_the_scenes_exact_types_runs_the_same_bits(hip_solver_factory, monkeypatch, tmp_path, plan):
    """bepuhip_specialise_units (round 6): the island kernel compiled — by hipcc on this box, from the library's own sources, on a thread of the library —
    for exactly the scene's types (two manifolds, BallSocket, and Weld and AngularServo of the widened set: the wide family's scene), cached in BEPUHIP_UNIT_CACHE. Before the
    unit is there the scene runs the all-44 family (2), afterwards the unit (family 3): every frame bit-exact against the oracle either way, and both contexts end with the
    same bits. A second context finds the object in the cache (no compiler run); another type set is another unit."""
    monkeypatch.setenv("BEPUHIP_UNIT_CACHE", str(tmp_path))
    types = [4, 7, 22, 31, 29]
    if plan == "split":
        monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "14")
        monkeypatch.setenv("BEPUHIP_FORCE_SPLIT", "64")
        scene = small_scenes.random_graph_scene(41, 2500, 7000, types)
    else:
        scene = small_scenes.island_scene(42, islands=60, bodies_per_island=14, constraints_per_island=40, type_ids=types)
    sd, cb = SolveDescription(2, 4), PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=4, threads=4)
    family = hip_solver_factory()
    got_family = pu.run_hip(family, scene, 1 / 60, sd, cb, frames=4)
    assert family.schedule() == (2 if plan == "split" else 1) and family.kernel_family() == 2
    _exact(ref, got_family)
    solver = hip_solver_factory()
    work = scene.copy()
    solver.upload(work, sd.fallback_batch_threshold)
    solver.solve(1 / 60, sd, cb)  # the unit is not there yet: the family's
    assert solver.kernel_family() == 2
    assert solver.specialise_units(wait=True) == 2, "hipcc is part of the image: the unit compiles"
    objects = sorted(p.name for p in tmp_path.iterdir() if p.suffix == ".so")
    assert len(objects) == 1 and objects[0].endswith("t512s.so" if plan == "split" else "t1024.so"), objects
    for _ in range(3):
        solver.solve(1 / 60, sd, cb)
    assert solver.kernel_family() == 3
    solver.download(work)
    _exact(ref, work)
    assert np.array_equal(work.bodies.view(np.int32), got_family.bodies.view(np.int32))
    again = hip_solver_factory()  # a second context: the cache has the object, and the registry of this process the loaded unit
    assert again.specialise_units(wait=True) == 0, "nothing uploaded yet: nothing to specialise"
    got_again = scene.copy()
    again.upload(got_again, sd.fallback_batch_threshold)  # ... but the context asks by itself from now on
    for _ in range(4):
        again.solve(1 / 60, sd, cb)
    assert again.specialise_units(wait=True) == 2 and again.kernel_family() in (2, 3)
    again.download(got_again)
    _exact(ref, got_again)
    assert sorted(p.name for p in tmp_path.iterdir() if p.suffix == ".so") == objects
