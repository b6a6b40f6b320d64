"""-m gpu: structural updates on the device: the narrow phase's per-frame add / remove stream applied to the rows in HBM
through bepuhip_add_constraint / remove_constraint, 60 frames with ~1 % of the contacts removed and as many added per frame, bit-exact against the oracle
solving the host mirror's export every frame — which is also what a fresh upload of that export would solve."""
import numpy as np
import pytest

import oracle_ffi
import parity_util as pu
import small_scenes
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks, SolveDescription
from mutable_scene import MutableSolver

pytestmark = pytest.mark.gpu

CONTACT_TYPES = [4, 5, 6, 7]
JOINT_TYPES = [22, 25, 30, 47]


def _build(seed, bodies=260, joints=300, contacts=500):
    rng = np.random.default_rng(seed)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-6, 6, 3)) if i % 37 else small_scenes.kinematic_body(rng, rng.uniform(-6, 6, 3)) for i in range(bodies)]
    ms = MutableSolver(np.stack(rows))

    def pair():
        while True:
            a, b = rng.choice(bodies, 2, replace=False)
            if not (ms.is_kinematic(a) and ms.is_kinematic(b)):
                return int(a), int(b)

    for _ in range(joints):
        a, b = pair()
        t = JOINT_TYPES[int(rng.integers(len(JOINT_TYPES)))]
        ms.add(t, [a, b], small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
    for _ in range(contacts):
        a, b = pair()
        t = CONTACT_TYPES[int(rng.integers(len(CONTACT_TYPES)))]
        ms.add(t, [a, b], small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
    return ms, rng, pair


@pytest.mark.parametrize("use_clusters", [False, True])
def test_sixty_frames_of_contact_churn_stay_bit_exact(hip_solver_factory, use_clusters):
    ms, rng, pair = _build(23)
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory(use_clusters=use_clusters)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    is_contact = lambda t: t in CONTACT_TYPES  # noqa: E731
    for frame in range(60):
        contacts = ms.locations(is_contact)
        churn = max(1, len(contacts) // 100)
        # removals: chosen one at a time (every swap-with-last renumbers the type batch)
        for _ in range(churn):
            locs = ms.locations(is_contact)
            bi, t, i = locs[int(rng.integers(len(locs)))]
            ms.remove(bi, t, i)
            solver.remove_constraint(bi, t, i)
        for _ in range(churn):
            a, b = pair()
            t = CONTACT_TYPES[int(rng.integers(len(CONTACT_TYPES)))]
            lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
            bi, index, encoded = ms.add(t, [a, b], lane)
            assert solver.add_constraint(bi, t, encoded, lane) == index
        export = ms.to_scene()
        for bi, tbs in enumerate(export.batches):
            for tb in tbs:
                assert solver.constraint_count(bi, tb.type_id) == tb.count
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        if frame % 12 == 11 or frame == 59:
            got = ms.to_scene()
            solver.download(got)
            m = pu.compare_scenes(export, got)
            assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    # whether the island schedule survived depends on where the random additions landed (see the two tests below); the results may not


@pytest.mark.parametrize("use_clusters", [False, True])
@pytest.mark.parametrize("mode", [1, 2])
def test_momentum_conserving_modes_after_structural_updates(hip_solver_factory, use_clusters, mode):
    """the conserving angular modes' substep-0 lists describe the topology, so structural updates used to end in UNSUPPORTED. They are now
    rebuilt from the rows on the device by the first conserving solve after a change: ten frames of contact churn, each solved in a conserving mode, bit-exact."""
    ms, rng, pair = _build(41, bodies=160, joints=180, contacts=260)
    sd, cb = SolveDescription(1, 3), PoseIntegratorCallbacks(angular_integration_mode=mode)
    solver = hip_solver_factory(use_clusters=use_clusters)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    is_contact = lambda t: t in CONTACT_TYPES  # noqa: E731
    for frame in range(10):
        for _ in range(4):
            locs = ms.locations(is_contact)
            bi, t, i = locs[int(rng.integers(len(locs)))]
            ms.remove(bi, t, i)
            solver.remove_constraint(bi, t, i)
            a, b = pair()
            t = CONTACT_TYPES[int(rng.integers(len(CONTACT_TYPES)))]
            lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
            bi, index, encoded = ms.add(t, [a, b], lane)
            assert solver.add_constraint(bi, t, encoded, lane) == index
        export = ms.to_scene()
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)


def test_addition_that_breaks_the_batch_invariant_is_refused_on_the_island_layout(hip_solver_factory):
    """a dynamic body may appear once per synchronized batch (Solver.cs:1046-1051; the reference asserts it in debug builds). On the island layout the host
    knows every reference, so an add_constraint into a batch that already holds one of its bodies is an INVALID_ARGUMENT instead of a silent race."""
    ms, rng, pair = _build(31, bodies=120, joints=140, contacts=200)
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene())
    locs = ms.locations(lambda t: t in CONTACT_TYPES)
    bi, t, i = locs[0]
    a, b = _decode(ms, ms.batches[bi][t]["refs"][i])
    dynamic = a if not ms.is_kinematic(a) else b
    other = next(x for x in range(ms.bodies.shape[0]) if x not in (a, b) and not ms.is_kinematic(x))
    lane = small_scenes.prestep_for(rng, t, ms.bodies[dynamic, 4:7], ms.bodies[other, 4:7])
    with pytest.raises(ValueError):
        solver.add_constraint(bi, t, np.asarray([dynamic, other], dtype=np.int32), lane)  # `dynamic` already has a constraint in batch bi
    sd, cb = SolveDescription(1, 2), PoseIntegratorCallbacks()
    solver.solve(1 / 60, sd, cb)  # nothing was changed: the context still solves the uploaded scene
    ref = ms.to_scene()
    oracle_ffi.solve(ref, 1 / 60, sd, cb)
    got = ms.to_scene()
    solver.download(got)
    m = pu.compare_scenes(ref, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"], m


def _lanes(tb, w, prestep):
    """Mask over an AOSOA buffer: True for the floats of occupied lanes (the trailing lanes of the last bundle hold nothing)."""
    fields = tb.prestep_floats if prestep else tb.impulse_floats
    bundles = (tb.count + w - 1) // w
    mask = np.zeros((bundles, fields, w), dtype=bool)
    idx = np.arange(tb.count)
    mask[idx // w, :, idx % w] = True
    return mask.reshape(-1)


def _decode(ms, encoded):
    from bepuphysics2_amd.scene import BODY_REFERENCE_MASK
    return [int(r) & BODY_REFERENCE_MASK for r in encoded]


def test_island_schedule_survives_the_refresh_of_persisting_pairs(hip_solver_factory):
    """What the narrow phase does to a pair whose manifold changed: remove the constraint, add one for the same bodies. On the island layout the removal frees a
    device slot where it is (the caller's indices are remapped: swap-with-last), the addition takes a free slot of its cluster's segment, the cluster's items fall back
    to batch-level waits — and the context stays on the island schedule, bit-exact against the oracle solving the host mirror every frame."""
    ms, rng, pair = _build(29)
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory()
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    is_contact = lambda t: t in CONTACT_TYPES  # noqa: E731
    for frame in range(30):
        for _ in range(8):
            locs = ms.locations(is_contact)
            bi, t, i = locs[int(rng.integers(len(locs)))]
            a, b = _decode(ms, ms.batches[bi][t]["refs"][i])
            ms.remove(bi, t, i)
            solver.remove_constraint(bi, t, i)
            lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
            bj, index, encoded = ms.add(t, [a, b], lane)  # first fit: the batch the pair just left
            assert solver.add_constraint(bj, t, encoded, lane) == index
        export = ms.to_scene()
        for bi, tbs in enumerate(export.batches):
            for tb in tbs:
                assert solver.constraint_count(bi, tb.type_id) == tb.count
        if frame % 4 == 1:
            # the narrow phase's in-place rewrites and the ranged read-backs address constraints by the caller's indices: they have to follow the remapping
            # (device index tables patched at the flush). Every contact type batch is rewritten with what the mirror holds — a no-op if, and only if, the map is right.
            for bi, tbs in enumerate(export.batches):
                for tb in tbs:
                    if tb.type_id in CONTACT_TYPES and tb.count:
                        bundles = (tb.count + export.bundle_width - 1) // export.bundle_width
                        assert np.array_equal(solver.get_prestep_range(bi, tb.type_id, 0, bundles).view(np.int32)[: tb.prestep.size][_lanes(tb, export.bundle_width, True)],
                                              tb.prestep.view(np.int32)[_lanes(tb, export.bundle_width, True)])
                        solver.update_prestep(bi, tb.type_id, 0, tb.prestep)
                        solver.update_accumulated_impulses(bi, tb.type_id, 0, tb.accumulated)
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        if frame % 6 == 5:
            got = ms.to_scene()
            solver.download(got)
            m = pu.compare_scenes(export, got)
            assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    assert solver.cluster_cycles().size > 0  # still the island schedule


def _degrees(ms):
    from bepuphysics2_amd.scene import BODY_REFERENCE_MASK, KINEMATIC_MASK
    deg = {}
    for b in ms.batches:
        for d in b.values():
            for refs in d["refs"]:
                for r in refs:
                    if not (r & KINEMATIC_MASK):
                        deg[r & BODY_REFERENCE_MASK] = deg.get(r & BODY_REFERENCE_MASK, 0) + 1
    return deg


def _first_fit_batch(ms, bodies):
    blocking = [b for b in bodies if not ms.is_kinematic(b)]
    for bi in range(len(ms.batches)):
        if not any(h in ms.batch_handles[bi] for h in blocking):
            return bi
    return len(ms.batches)


def test_reserved_slots_take_new_contacts_inside_an_island(hip_solver_factory):
    """BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS: every cluster segment is planned with spare device slots. New contacts between bodies of the island (of a manifold type the
    batch they fall into already holds) and removals that leave no body bare go on for twenty frames without leaving the island schedule."""
    ms, rng, pair = _build(37, bodies=200, joints=260, contacts=420)
    sd, cb = SolveDescription(2, 3), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    is_contact = lambda t: t in CONTACT_TYPES  # noqa: E731
    added = removed = 0
    for frame in range(20):
        deg = _degrees(ms)
        for _ in range(3):
            locs = [(bi, t, i) for bi, t, i in ms.locations(is_contact) if all(ms.is_kinematic(b) or deg.get(b, 0) > 1 for b in _decode(ms, ms.batches[bi][t]["refs"][i]))]
            bi, t, i = locs[int(rng.integers(len(locs)))]
            for b in _decode(ms, ms.batches[bi][t]["refs"][i]):
                deg[b] = deg.get(b, 0) - 1
            ms.remove(bi, t, i)
            solver.remove_constraint(bi, t, i)
            removed += 1
        for _ in range(40):
            if added >= 2 * (frame + 1):
                break
            a, b = pair()
            if ms.is_kinematic(a) or ms.is_kinematic(b) or deg.get(a, 0) < 1 or deg.get(b, 0) < 1:  # a body without constraints is not part of the plan
                continue
            bi = _first_fit_batch(ms, [a, b])
            present = [t for t in CONTACT_TYPES if bi < len(ms.batches) and t in ms.batches[bi] and len(ms.batches[bi][t]["refs"]) > 0]
            if not present:
                continue
            t = present[int(rng.integers(len(present)))]
            lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
            bj, index, encoded = ms.add(t, [a, b], lane)
            assert bj == bi and solver.add_constraint(bj, t, encoded, lane) == index
            added += 1
        export = ms.to_scene()
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    assert added >= 30 and removed == 60
    assert solver.cluster_cycles().size > 0  # never left the island schedule


def test_new_batches_type_batches_and_capacity_growth(hip_solver_factory):
    """A hub that collects constraint after constraint opens new batches and new type batches and overflows the 64-lane rows of existing ones."""
    rng = np.random.default_rng(5)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-3, 3, 3)) for _ in range(140)]
    ms = MutableSolver(np.stack(rows))
    ms.add(7, [1, 2], small_scenes.prestep_for(rng, 7, ms.bodies[1, 4:7], ms.bodies[2, 4:7]))
    sd, cb = SolveDescription(2, 2), PoseIntegratorCallbacks()
    solver = hip_solver_factory()
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    for k in range(3, 40):     # hub 0: every constraint needs a batch of its own
        t = [7, 22, 4][k % 3]
        lane = small_scenes.prestep_for(rng, t, ms.bodies[0, 4:7], ms.bodies[k, 4:7])
        bi, index, encoded = ms.add(t, [0, k], lane)
        assert solver.add_constraint(bi, t, encoded, lane) == index
    for k in range(40, 139, 2):  # disjoint pairs: all land in batch 0 and overflow its Contact4 rows (64 -> 128)
        lane = small_scenes.prestep_for(rng, 7, ms.bodies[k, 4:7], ms.bodies[k + 1, 4:7])
        bi, index, encoded = ms.add(7, [k, k + 1], lane)
        assert solver.add_constraint(bi, 7, encoded, lane) == index
    for _ in range(3):
        export = ms.to_scene()
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
    got = ms.to_scene()
    solver.download(got)
    m = pu.compare_scenes(export, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
    # a body memory move: body 139 takes the slot of body 3 (BodySet.RemoveAt swap-with-last); its constraints' references are patched one by one
    with pytest.raises(ValueError):
        solver.remove_constraint(0, 7, 10_000)


@pytest.mark.parametrize("reserve", [False, True])
def test_structural_updates_after_frames_on_the_split_island_plan(hip_solver_factory, monkeypatch, reserve):
    """One island too large for a workgroup: the first frames run the split-island plan (rows permuted per cluster, rank rows behind the local references), then the
    narrow phase starts adding and removing random pairs — most of them across clusters. Round 3: with reserved slots the plan absorbs them (the
    remote body becomes a shared body with a ghost slot in the cluster that runs the constraint, the touched bodies are re-ranked, the clusters' predecessor lists
    rebuilt) and the context STAYS on the island schedule; without the reserve it stays as long as removals have left room and otherwise brings the rows back into the
    caller's order without losing what the frames accumulated. Bit-exact against the oracle every frame either way."""
    monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "12")
    ms, rng, pair = _build(31, bodies=2600, joints=3000, contacts=5000)
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=reserve)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    is_contact = lambda t: t in CONTACT_TYPES  # noqa: E731
    for frame in range(8):
        if frame >= 3:
            for _ in range(20):
                locs = ms.locations(is_contact)
                bi, t, i = locs[int(rng.integers(len(locs)))]
                if reserve and any(not (r & 0x40000000) and ms.dynamic_degree(r) < 2 for r in ms.batches[bi][t]["refs"][i]):
                    continue  # (a body that loses its last constraint leaves ANY island plan: the reference integrates it as an unconstrained body from then on)
                ms.remove(bi, t, i)
                solver.remove_constraint(bi, t, i)
            for _ in range(20):
                a, b = pair()
                t = CONTACT_TYPES[int(rng.integers(len(CONTACT_TYPES)))]
                lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
                bi, index, encoded = ms.add(t, [a, b], lane)
                assert solver.add_constraint(bi, t, encoded, lane) == index
        export = ms.to_scene()
        oracle_ffi.solve(export, 1 / 60, sd, cb, threads=4)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        if frame == 2:
            assert solver.cluster_cycles().size > 1  # the split plan ran the first frames
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    if reserve:
        assert solver.cluster_cycles().size > 1, "structural updates with reserved slots must keep the split-island plan"


@pytest.mark.parametrize("mode", [0, 1])
def test_cpp_simulation_adds_and_removes_between_timesteps(hip_solver_factory, mode):
    """The C++ mirror end to end: Simulation + Solver.Add / Solver.Remove between Timestep calls, HipTimestepper bringing the device up to date without a re-upload —
    mode 0 by replaying the solver's structural log (what a listener inside the reference would record) through bepuhip_add_constraint / remove_constraint, mode 1 by
    diffing the type batches' handles and references against last frame's copy (public API only: TypeBatch.IndexToHandle, TypeBatch.cs:16) and sending the frame's
    changes in ONE bepuhip_apply_structural_ops call. Ragdolls in a tube: every frame a few contacts of every ragdoll-vs-tube manifold type are removed and
    contacts of the same bodies come back (the narrow phase's refresh); each frame's result equals the oracle's solve of that frame's export."""
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene("ragdoll_tube", 80, 1, 0, 9)
    sd, cb = sim.solve_description(), PoseIntegratorCallbacks()
    sim.attach_hip_timestepper(0)
    sim.timestepper_mode(mode)
    rng = np.random.default_rng(4)
    contact_ids = {t for t, info in small_scenes.TYPE_TABLE.items() if info[3].startswith("Contact") and "Nonconvex" not in info[3]}
    for frame in range(12):
        if frame >= 2:
            before = sim.export()
            handles = sim.constraint_handles(lambda t: t in contact_ids)
            for h in rng.choice(handles, size=12, replace=False):
                # find the constraint's lane in the export (handle -> location is the mirror's business; the test reads it back from the type batch handles)
                found = None
                for bi, tbs in enumerate(before.batches):
                    for ti, tb in enumerate(tbs):
                        if tb.type_id in contact_ids and tb.count:
                            hs = np.ctypeslib.as_array(sim.lib.bepuhost_type_batch_handles(sim.h, bi, ti), shape=(tb.count,))
                            hit = np.nonzero(hs == h)[0]
                            if hit.size:
                                found = (tb.type_id, tb.refs_lanes(8)[int(hit[0])].copy(), tb.prestep_lanes(8)[int(hit[0])].copy())
                    if found:
                        break
                assert found is not None
                type_id, refs, lane = found
                sim.remove_constraint(int(h))
                bodies = [int(before.index_to_handle[int(r) & 0x3FFFFFFF]) for r in refs]
                sim.add_constraint(type_id, bodies, lane)
                before = sim.export()
            sim.validate()
        export = sim.export()
        ref = export.copy()
        oracle_ffi.solve(ref, 1 / 60, sd, cb, threads=4)
        sim.timestep(1 / 60)
        got = sim.export()
        m = pu.compare_scenes(ref, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    uploads, replays = sim.timestepper_stats()
    assert uploads == 1 and replays == 10, (uploads, replays)
    sim.close()


def test_reserved_layout_falls_back_cleanly_when_the_island_schedule_does_not_apply(hip_solver_factory, monkeypatch):
    """A context planned with spare device slots is asked — with BEPUHIP_CONSERVING_CLUSTERS=0, round 2's behaviour — for a momentum-conserving angular mode: the
    launch-per-batch kernels address rows [0, count), so the rows first go back into the caller's order — results as always. (By default the island schedule runs the
    conserving modes itself: third case; and more substeps than one launch carries are a chain of island launches since round 5: first case, no fall-back any more.)"""
    scene = small_scenes.island_scene(5, 30, 12, 28, [22, 4, 30, 47, 7, 5])
    for sd, cb, conserving_clusters in ((SolveDescription(1, 65), PoseIntegratorCallbacks(), "1"), (SolveDescription(2, 3), PoseIntegratorCallbacks(angular_integration_mode=1), "0"),
                                        (SolveDescription(2, 3), PoseIntegratorCallbacks(angular_integration_mode=1), "1")):
        monkeypatch.setenv("BEPUHIP_CONSERVING_CLUSTERS", conserving_clusters)
        ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=2)
        solver = hip_solver_factory(reserve_update_slots=True)
        got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=2)
        assert (solver.cluster_cycles().size == 0) == (conserving_clusters == "0")
        m = pu.compare_scenes(ref, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def test_replan_puts_a_context_that_left_its_plan_back_on_the_island_schedule(hip_solver_factory):
    """bepuhip_replan (round 3): the hub scene of the test above drops the context to launch-per-batch (new batches, new type batches, rows that grow); a re-plan reads the
    references back, plans the constraints the device holds, permutes prestep data and accumulated impulses on the device — and the following frames run on the island
    schedule with the same bits as the oracle. reset_state after the re-plan returns to the state of the last structural update, in the new layout."""
    rng = np.random.default_rng(5)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-3, 3, 3)) for _ in range(140)]
    ms, pristine = MutableSolver(np.stack(rows)), MutableSolver(np.stack(rows))  # `pristine` gets every structural update and no frame: what reset_state returns to
    for k in range(1, 139, 2):
        lane = small_scenes.prestep_for(rng, 7, ms.bodies[k, 4:7], ms.bodies[k + 1, 4:7])
        ms.add(7, [k, k + 1], lane); pristine.add(7, [k, k + 1], lane)
    sd, cb = SolveDescription(2, 2), PoseIntegratorCallbacks()
    solver = hip_solver_factory()
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    assert solver.schedule() == 1

    def frames(n):
        for _ in range(n):
            export = ms.to_scene()
            oracle_ffi.solve(export, 1 / 60, sd, cb)
            ms.absorb(export)
            solver.solve(1 / 60, sd, cb)
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m

    frames(2)
    for k in range(3, 40):  # hub 0 had no constraint, and every constraint on it needs a batch of its own: nothing an island plan absorbs
        t = [7, 22, 4][k % 3]
        lane = small_scenes.prestep_for(rng, t, ms.bodies[0, 4:7], ms.bodies[k, 4:7])
        bi, index, encoded = ms.add(t, [0, k], lane)
        pristine.add(t, [0, k], lane)
        assert solver.add_constraint(bi, t, encoded, lane) == index
    frames(2)
    assert solver.schedule() == 0 and solver.cluster_cycles().size == 0
    solver.replan()
    assert solver.schedule() == 1
    for bi, tbs in enumerate(ms.to_scene().batches):
        for tb in tbs:
            assert solver.constraint_count(bi, tb.type_id) == tb.count
    before = ms.to_scene()
    solver.download(before)  # read-backs in the caller's order see the same values through the new layout
    m = pu.compare_scenes(ms.to_scene(), before)
    assert m["impulses_bit_exact"] and m["prestep_bit_exact"], m
    frames(3)
    assert solver.cluster_cycles().size > 0
    # a second re-plan of a context that IS on a plan (fresh reserves) changes nothing either
    solver.replan()
    frames(2)
    # the snapshot moved into the new layout with its own values: uploaded bodies and rows + the structural updates, no frame
    solver.reset_state()
    solver.solve(1 / 60, sd, cb)
    want = pristine.to_scene()
    oracle_ffi.solve(want, 1 / 60, sd, cb)
    got = pristine.to_scene()
    solver.download(got)
    m = pu.compare_scenes(want, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def test_replan_on_a_split_island_plan_and_the_snapshot_reset_state_returns_to(hip_solver_factory, monkeypatch):
    monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "12")
    ms, rng, pair = _build(37, bodies=2600, joints=3000, contacts=5000)
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory()  # no reserved slots: the first additions across clusters leave the plan
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    assert solver.schedule() == 2
    for frame in range(6):
        if frame in (1, 2):
            for _ in range(40):
                a, b = pair()
                t = CONTACT_TYPES[int(rng.integers(len(CONTACT_TYPES)))]
                lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
                bi, index, encoded = ms.add(t, [a, b], lane)
                assert solver.add_constraint(bi, t, encoded, lane) == index
        if frame == 3:
            assert solver.schedule() == 0
            solver.replan()
            assert solver.schedule() == 2
        export = ms.to_scene()
        oracle_ffi.solve(export, 1 / 60, sd, cb, threads=4)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    assert solver.cluster_cycles().size > 1


def test_a_kinematic_body_that_loses_and_regains_its_constraints_on_the_island_layout(hip_solver_factory):
    """Solver.ConstrainedKinematicHandles follows the constraints (Solver.cs:1025, 1368-1377): a kinematic body whose last constraint goes is integrated as an
    unconstrained body from then on (once per frame, not per substep), and per substep again when a constraint comes back. On the island layout the plan's own list of
    constrained kinematic bodies has to follow (bepu_soft_updates.h, kin_uses) — the context stays on the island schedule throughout."""
    rng = np.random.default_rng(77)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-3, 3, 3)) for _ in range(60)] + [small_scenes.kinematic_body(rng, rng.uniform(-3, 3, 3), angular=(0.3, 0.1, 0.25)) for _ in range(2)]
    ms = MutableSolver(np.stack(rows))
    for k in range(0, 58, 2):
        ms.add(7, [k, k + 1], small_scenes.prestep_for(rng, 7, ms.bodies[k, 4:7], ms.bodies[k + 1, 4:7]))
    for k, kin in ((3, 60), (8, 60), (20, 61)):
        ms.add(5, [k, kin], small_scenes.prestep_for(rng, 5, ms.bodies[k, 4:7], ms.bodies[kin, 4:7]))
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)

    def frames(n):
        for _ in range(n):
            export = ms.to_scene()
            kin = np.ascontiguousarray(export.constrained_kinematic_indices(), dtype=np.int32)  # the caller re-sends the set when it changes
            solver.set_constrained_kinematics(kin)
            oracle_ffi.solve(export, 1 / 60, sd, cb)
            ms.absorb(export)
            solver.solve(1 / 60, sd, cb)
            got = ms.to_scene()
            solver.download(got)
            m = pu.compare_scenes(export, got)
            assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
        assert solver.schedule() == 1

    frames(2)
    loc = [(bi, t, i) for bi, t, i in ms.locations(lambda t: t == 5) if (ms.batches[bi][t]["refs"][i][1] & 0x3FFFFFFF) == 61]
    assert len(loc) == 1
    lane = list(ms.batches[loc[0][0]][5]["prestep"][loc[0][2]])
    ms.remove(*loc[0])
    solver.remove_constraint(*loc[0])
    assert 61 not in ms.kinematic_constrained
    frames(3)  # body 61 is an unconstrained kinematic body now
    bi, index, encoded = ms.add(5, [20, 61], lane)
    assert solver.add_constraint(bi, 5, encoded, lane) == index
    frames(3)


@pytest.mark.parametrize("use_clusters", [False, True])
def test_body_removal_moves_the_last_body_and_patches_its_references(hip_solver_factory, use_clusters):
    """Bodies.RemoveAt (BodySet.cs:83-110): the last body takes the removed body's slot and Solver.UpdateForBodyMemoryMove patches every constraint that referenced it
    (bepuhip_update_body_reference), the body array shrinks (bepuhip_set_bodies). Dynamic and kinematic bodies; on an island plan the moved body keeps its cluster,
    its LDS slot and its constraints under the new index (soft_move_body) and the context stays on the plan. Bit-exact against the oracle solving the host mirror every
    frame."""
    ms, rng, pair = _build(41, bodies=200, joints=220, contacts=380)
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory(use_clusters=use_clusters, reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)

    def frames(n):
        for _ in range(n):
            export = ms.to_scene()
            solver.set_constrained_kinematics(export.constrained_kinematic_indices())
            oracle_ffi.solve(export, 1 / 60, sd, cb)
            ms.absorb(export)
            solver.solve(1 / 60, sd, cb)
            got = ms.to_scene()
            solver.download(got)
            m = pu.compare_scenes(export, got)
            assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m

    frames(2)
    for victim in (5, 37, 0):  # a dynamic body, a kinematic one (every 37th), and one whose slot the then-last body (whatever it is) takes
        for bi, t, i in sorted(((bi, t, i) for bi, t, i in ms.locations() if any((int(r) & 0x3FFFFFFF) == victim for r in ms.batches[bi][t]["refs"][i])), reverse=True):
            ms.remove(bi, t, i)
            solver.remove_constraint(bi, t, i)
        for bi, t, i, k, encoded in ms.remove_body(victim):
            solver.update_body_reference(bi, t, i, k, encoded)
        solver.set_bodies(ms.bodies)
        frames(2)
        if use_clusters:
            assert solver.schedule() in (1, 2)  # the moved body keeps its place in the plan under its new index; the removed one left it with its last constraint


@pytest.mark.parametrize("split", [False, True])
def test_bodies_join_and_leave_the_plan_with_their_first_and_last_constraint(hip_solver_factory, monkeypatch, split):
    """A body without constraints is not part of an island plan (the tail integrates it once per frame, as the reference integrates unconstrained bodies). Its first
    constraint brings it into the cluster that runs the constraint — an unused LDS slot, an entry in the list behind kFlagClustered — and its last one takes it out again
    (round 3; round 2 left the island schedule in both cases). Two unconstrained bodies that get a constraint between them form a new island in a cluster with room.
    The context stays on its plan throughout; bit-exact against the oracle every frame."""
    if split:
        monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "12")
    rng = np.random.default_rng(9)
    connected = 2600 if split else 60
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-3, 3, 3)) for _ in range(connected + 20)]
    ms = MutableSolver(np.stack(rows))
    for k in range(connected - 1):
        ms.add(7 if k % 2 else 5, [k, k + 1], small_scenes.prestep_for(rng, 7 if k % 2 else 5, ms.bodies[k, 4:7], ms.bodies[k + 1, 4:7]))
    for _ in range(connected if split else 0):  # (a chain alone would be cut into strips with hardly any shared body)
        a, b = (int(x) for x in rng.choice(connected, 2, replace=False))
        ms.add(4, [a, b], small_scenes.prestep_for(rng, 4, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    want = 2 if split else 1
    assert solver.schedule() == want

    def frames(n):
        for _ in range(n):
            export = ms.to_scene()
            oracle_ffi.solve(export, 1 / 60, sd, cb, threads=4)
            ms.absorb(export)
            solver.solve(1 / 60, sd, cb)
            got = ms.to_scene()
            solver.download(got)
            m = pu.compare_scenes(export, got)
            assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m
        assert solver.schedule() == want

    def add(a, b):  # a contact between a and b, of a type its batch (Solver.Add's first fit) already holds: nothing here opens a batch or a type batch
        bi = next(i for i in range(len(ms.batches)) if a not in ms.batch_handles[i] and b not in ms.batch_handles[i])
        t = ms.type_order[bi][0]
        lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
        bi, index, encoded = ms.add(t, [a, b], lane)
        assert solver.add_constraint(bi, t, encoded, lane) == index
        return bi, t, index

    frames(2)
    free = connected  # the first body without constraints
    add(free, free + 1)                        # two newcomers: a new island
    joined = add(0, free + 2)                  # a newcomer joins body 0's cluster
    frames(2)
    ms.remove(*joined); solver.remove_constraint(*joined)      # ... and leaves again
    add(free + 1, free + 3)                    # the new island grows by another newcomer
    frames(2)
    for location in sorted(ms.locations(lambda t: True), reverse=True):  # the new island dissolves: three bodies leave
        if any((int(r) & 0x3FFFFFFF) >= free for r in ms.batches[location[0]][location[1]]["refs"][location[2]]):
            ms.remove(*location); solver.remove_constraint(*location)
    frames(2)
    add(free, connected - 1)                   # and one of them comes back
    frames(2)


def _diff_ops(before, ms):
    """The operations that bring a device holding `before` (MutableSolver.snapshot) to the solver's present state: hostlib.diff_type_batch per type batch (the C++ twin of
    the C# shim's diff), concatenated into one table for bepuhip_apply_structural_ops."""
    from bepuphysics2_amd import hostlib
    scene = ms.to_scene()
    tables, payloads, words = [], [], 0
    seen = set()
    for bi, tbs in enumerate(scene.batches):
        for tb, t in zip(tbs, ms.type_order[bi]):
            seen.add((bi, t))
            nb, pf, _imf, _ = TYPE_TABLE[t]
            old_handles, old_refs = before.get((bi, t), (np.zeros(0, np.int32), np.zeros((0, nb), np.int32)))
            handles = np.asarray(ms.batches[bi][t]["handles"], dtype=np.int32)
            if tb.count == 0 and old_handles.size == 0:
                continue
            refs = tb.body_refs if tb.count else np.zeros(nb * 8, np.int32)
            pre = tb.prestep if tb.count else np.zeros(pf * 8, np.float32)
            ops, payload = hostlib.diff_type_batch(bi, t, nb, pf, old_handles, old_refs, handles, refs, pre)
            if ops.shape[0]:
                ops[ops[:, 0] == 0, 6] += words
                used = int(np.count_nonzero(ops[:, 0] == 0)) * (nb + pf)
                tables.append(ops); payloads.append(payload[:used]); words += used
    for (bi, t), (old_handles, _) in before.items():
        assert (bi, t) in seen or old_handles.size == 0
    table = np.concatenate(tables) if tables else np.zeros((0, 8), np.int32)
    payload = np.concatenate(payloads) if payloads and words else np.zeros(1, np.uint32)
    return table, payload


@pytest.mark.parametrize("layout", ["whole_islands", "split_plan", "launch_per_batch"])
def test_a_frames_changes_reconstructed_from_the_type_batches_and_sent_in_one_call(hip_solver_factory, monkeypatch, layout):
    """the device is kept up to date by ONE bepuhip_apply_structural_ops call per frame whose operations are not recorded but RECONSTRUCTED — the
    diff of every type batch's constraint handles and body references against last frame's copy (what integration/csharp/HipTimestepper.cs does on the unpatched
    reference). Frames of contact churn, of survivors rearranged (the reference removed in another order than the diff assumes: swaps), of a body removal (the last body
    moves into the freed index: reference patches); every frame equal to the oracle solving the host mirror."""
    rng = np.random.default_rng(23)
    if layout == "split_plan":
        monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "12")
        nb, nc = 2400, 7000
    else:
        nb, nc = 260, 700
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-6, 6, 3)) if i % 19 else small_scenes.kinematic_body(rng, rng.uniform(-6, 6, 3)) for i in range(nb)]
    ms = MutableSolver(np.stack(rows))
    types = [4, 5, 6, 7, 22, 25, 47, 0, 3]

    def add_random():
        t = types[int(rng.integers(len(types)))]
        one_body = TYPE_TABLE[t][0] == 1
        while True:
            a, b = (int(x) for x in rng.choice(ms.bodies.shape[0], 2, replace=False))
            if one_body and not ms.is_kinematic(a):
                bodies = [a]
                break
            if not one_body and not (ms.is_kinematic(a) and ms.is_kinematic(b)):
                bodies = [a, b]
                break
        ms.add(t, bodies, small_scenes.prestep_for(rng, t, ms.bodies[bodies[0], 4:7], ms.bodies[bodies[-1], 4:7]))

    if layout == "split_plan":  # neighbours only: one big island with a sensible cut
        for i in range(nc):
            a = int(rng.integers(nb - 12))
            b = a + int(rng.integers(1, 12))
            if ms.is_kinematic(a) and ms.is_kinematic(b):
                continue
            t = [4, 5, 6, 7, 22, 25, 47][int(rng.integers(7))]
            ms.add(t, [a, b], small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
    else:
        for _ in range(nc):
            add_random()
    sd, cb = SolveDescription(2, 3), PoseIntegratorCallbacks()
    solver = hip_solver_factory(use_clusters=layout != "launch_per_batch", reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    sent = 0
    for frame in range(8):
        before = ms.snapshot()
        for _ in range(int(rng.integers(4, 14))):
            locs = ms.locations()
            bi, t, i = locs[int(rng.integers(len(locs)))]
            ms.remove(bi, t, i)
        for _ in range(int(rng.integers(4, 14))):
            if layout != "split_plan":
                add_random()
                continue
            a = int(rng.integers(ms.bodies.shape[0] - 12))  # a split plan's narrow phase: new contacts between neighbours
            b = a + int(rng.integers(1, 12))
            if not (ms.is_kinematic(a) and ms.is_kinematic(b)):
                t = [4, 5, 6, 7][int(rng.integers(4))]
                ms.add(t, [a, b], small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
        for _ in range(3):  # survivors in another arrangement than remove-then-append leaves them in
            locs = ms.locations()
            bi, t, i = locs[int(rng.integers(len(locs)))]
            n = len(ms.batches[bi][t]["refs"])
            if n > 1:
                ms.swap(bi, t, i, int(rng.integers(n)))
        bodies_changed = False
        if frame in (3, 6):  # Bodies.RemoveAt: a body loses its constraints, the last body takes its index
            victim = int(rng.integers(ms.bodies.shape[0] - 1))
            for bi, t, i in sorted((loc for loc in ms.locations() if any((int(r) & 0x3FFFFFFF) == victim for r in ms.batches[loc[0]][loc[1]]["refs"][loc[2]])), reverse=True):
                ms.remove(bi, t, i)
            ms.remove_body(victim)
            bodies_changed = True
        table, payload = _diff_ops(before, ms)
        sent += table.shape[0]
        solver.apply_structural_op_table(table, payload)
        if bodies_changed:
            solver.set_bodies(ms.bodies)
        export = ms.to_scene()
        solver.set_constrained_kinematics(export.constrained_kinematic_indices())
        oracle_ffi.solve(export, 1 / 60, sd, cb, threads=4)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (layout, frame, m)
    assert sent > 100
    if layout == "launch_per_batch":
        assert solver.schedule() == 0


def test_batched_operations_stop_at_the_first_failure_and_say_which(hip_solver_factory):
    """bepuhip_apply_structural_ops applies its table in order, each operation exactly as the single call: the first failure ends it, is reported with its ordinal,
    and what came before stays applied (the device equals the oracle on the mirror that took the same operations)."""
    rng = np.random.default_rng(2)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-4, 4, 3)) for _ in range(120)]
    ms = MutableSolver(np.stack(rows))
    for _ in range(300):
        a, b = (int(x) for x in rng.choice(120, 2, replace=False))
        t = [4, 7, 22, 47][int(rng.integers(4))]
        ms.add(t, [a, b], small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7]))
    sd, cb = SolveDescription(1, 2), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    bi, t, _ = ms.locations()[5]
    count = len(ms.batches[bi][t]["refs"])
    lane = small_scenes.prestep_for(rng, t, ms.bodies[0, 4:7], ms.bodies[1, 4:7])
    with pytest.raises(Exception, match="structural operation 1 of 3"):
        solver.apply_structural_ops([("remove", bi, t, 0), ("remove", bi, t, count + 7), ("add", bi, t, [0, 1], lane)])
    ms.remove(bi, t, 0)  # the first operation happened, the third did not
    assert solver.constraint_count(bi, t) == count - 1
    solver.apply_structural_ops([("swap", bi, t, 0, 1)])
    ms.swap(bi, t, 0, 1)
    export = ms.to_scene()
    oracle_ffi.solve(export, 1 / 60, sd, cb)
    solver.solve(1 / 60, sd, cb)
    got = ms.to_scene()
    solver.download(got)
    m = pu.compare_scenes(export, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def test_structural_updates_beside_a_sequential_fallback_batch(hip_solver_factory):
    """Round 5 (VERDICT r4 missing #6): a scene with a sequential fallback batch used to refuse every structural update (UNSUPPORTED -> a full upload). Constraints of
    the synchronized batches now come and go in place — the context leaves its island layout for the first update, the fallback batch's dependency levels are rebuilt
    from its rows at the flush —; the fallback batch's own rows stay what was uploaded: removals from it are still refused. Every frame a constraint of a synchronized
    batch is removed (swap-with-last) and the same pair comes back at the end of its type batch with a new prestep lane and zero impulses; bit for bit the oracle."""
    from bepuphysics2_amd.native import UnsupportedError
    from bepuphysics2_amd.scene import KINEMATIC_MASK, to_aosoa
    threshold = 5
    scene = small_scenes.star_scene(6, spokes=40, hubs=2, fallback_batch_threshold=threshold)
    assert len(scene.batches) == threshold + 1
    sd, cb = SolveDescription(1, 3, fallback_batch_threshold=threshold), PoseIntegratorCallbacks()
    solver = hip_solver_factory()
    solver.upload(scene, threshold)
    solver.solve(1 / 60, sd, cb)
    oracle_ffi.solve(scene, 1 / 60, sd, cb)
    rng = np.random.default_rng(5)
    w = scene.bundle_width
    with pytest.raises(UnsupportedError):  # (round 6: the fallback batch takes removals and placed additions — the test below; swaps inside it stay refused, and appends go through add_constraint_at)
        solver.swap_constraints(threshold, scene.batches[threshold][0].type_id, 0, 1)
    with pytest.raises(UnsupportedError):
        tb0 = scene.batches[threshold][0]
        solver.add_constraint(threshold, tb0.type_id, [int(r) for r in tb0.refs_lanes(scene.bundle_width)[0]], tb0.prestep_lanes(scene.bundle_width)[0])
    for frame in range(8):
        candidates = [(bi, k) for bi in range(threshold) for k, tb in enumerate(scene.batches[bi]) if tb.count >= 2]
        bi, k = candidates[int(rng.integers(len(candidates)))]
        tb = scene.batches[bi][k]
        t, n = tb.type_id, tb.count
        refs, pre, acc = tb.refs_lanes(w), tb.prestep_lanes(w), tb.accumulated_lanes(w)
        i = int(rng.integers(n))
        pair = refs[i].copy()
        solver.remove_constraint(bi, t, i)
        refs[i], pre[i], acc[i] = refs[n - 1], pre[n - 1], acc[n - 1]  # TypeProcessor.Remove: the last constraint takes the index
        position = [scene.bodies[int(r) & ~KINEMATIC_MASK, 4:7] for r in pair]
        lane = np.asarray(small_scenes.prestep_for(rng, t, position[0], position[1] if len(position) > 1 else None), dtype=np.float32)
        assert solver.add_constraint(bi, t, [int(r) for r in pair], lane) == n - 1
        refs[n - 1], pre[n - 1], acc[n - 1] = pair, lane, 0.0
        tb.body_refs, tb.prestep, tb.accumulated = to_aosoa(refs.astype(np.int32), w, fill=-1), to_aosoa(pre.astype(np.float32), w), to_aosoa(acc.astype(np.float32), w)
        export = scene.copy()
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        solver.solve(1 / 60, sd, cb)
        got = scene.copy()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, bi, t, m)
        scene = export


@pytest.mark.parametrize("use_clusters", [False, True])
def test_the_sequential_fallback_batch_itself_takes_additions_and_removals(hip_solver_factory, use_clusters):
    """Round 6 (VERDICT r5 missing #4): structural changes OF the fallback batch — legal in the reference (TypeProcessor.cs:451-571 allocation by probing, :633-694 removal
    with bundle compaction, Solver.cs:59,202,236) — used to be UNSUPPORTED -> an 8 ms re-upload. The host mirror (tests/mutable_scene.py) applies the reference's rules,
    handle-hashed probing included; the device follows through bepuhip_add_constraint_at (the lane is the reference's choice) and bepuhip_remove_constraint, and every
    frame is compared with the oracle solving the mirror, bit for bit: hub bodies gain and lose constraints for 24 frames, type batches of the fallback batch grow past
    17 bundles (hashed probes), shrink through bundle moves, empty out and come back; one frame goes through the batched entry point (kind 4)."""
    from mutable_scene import MutableSolver
    from bepuphysics2_amd.scene import KINEMATIC_MASK
    threshold = 4
    rng = np.random.default_rng(17)
    hubs, spokes = 3, 100
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-1, 1, 3)) for _ in range(hubs)]
    rows += [small_scenes.random_dynamic_body(rng, rng.uniform(-4, 4, 3)) if i % 9 else small_scenes.kinematic_body(rng, rng.uniform(-4, 4, 3)) for i in range(spokes)]
    ms = MutableSolver(np.stack(rows), fallback_batch_threshold=threshold)
    types = (7, 22, 4, 47, 30)

    def add_spoke(solver=None, ops=None, payload=None):
        t = types[int(rng.integers(len(types)))]
        hub, spoke = int(rng.integers(hubs)), hubs + int(rng.integers(spokes))
        pair = [hub, spoke] if rng.integers(2) else [spoke, hub]
        lane = np.asarray(small_scenes.prestep_for(rng, t, ms.bodies[pair[0], 4:7], ms.bodies[pair[1], 4:7]), dtype=np.float32)
        bi, index, encoded = ms.add(t, pair, lane)
        if ops is not None:
            ops.append((4 if bi == threshold else 0, bi, t, index, 0, 0, len(payload)))
            payload.extend(int(np.int32(e).view(np.uint32)) if e < 0 else int(e) for e in encoded)
            payload.extend(int(w) for w in lane.view(np.uint32))
        elif solver is not None:
            if bi == threshold:
                solver.add_constraint_at(bi, t, index, encoded, lane)
            else:
                assert solver.add_constraint(bi, t, encoded, lane) == index

    for _ in range(150):
        add_spoke()
    assert len(ms.batches) == threshold + 1
    sd, cb = SolveDescription(1, 3, fallback_batch_threshold=threshold), PoseIntegratorCallbacks()
    solver = hip_solver_factory(use_clusters=use_clusters)
    solver.upload(ms.to_scene(), threshold)
    grew_past_17 = emptied = False
    for frame in range(24):
        fallback = [loc for loc in ms.locations() if loc[0] == threshold]
        shrink = frame in (9, 10, 11, 12, 13) or (frame % 2 == 1 and len(fallback) > 40)
        if frame == 16:  # one frame through bepuhip_apply_structural_ops
            ops, payload = [], []
            for _ in range(12):
                add_spoke(ops=ops, payload=payload)
            table = np.zeros((len(ops), 8), dtype=np.int32)
            table[:, :7] = np.asarray(ops, dtype=np.int64).astype(np.int32)
            solver.apply_structural_op_table(table, np.asarray(payload, dtype=np.uint32))
        elif shrink:
            for _ in range(min(len(fallback), 45 if frame in (9, 10, 11, 12, 13) else 14)):
                fallback = [loc for loc in ms.locations() if loc[0] == threshold]
                if frame == 9:  # one type batch of the fallback batch loses everything (and gets constraints again later)
                    fallback = [loc for loc in fallback if loc[1] == 30]
                if not fallback:
                    break
                bi, t, i = fallback[int(rng.integers(len(fallback)))]
                ms.remove(bi, t, i)
                solver.remove_constraint(bi, t, i)
        else:
            for _ in range(40 if frame < 6 else 16):
                add_spoke(solver)
        if len(ms.batches) > threshold:
            grew_past_17 |= any(len(tb["refs"]) > 17 * ms.w for tb in ms.batches[threshold].values())
            emptied |= any(len(tb["refs"]) == 0 for tb in ms.batches[threshold].values())
        export = ms.to_scene()
        for bi, tbs in enumerate(export.batches):
            for tb in tbs:
                assert solver.constraint_count(bi, tb.type_id) == tb.count, (frame, bi, tb.type_id)
        kin = np.ascontiguousarray(export.constrained_kinematic_indices(), dtype=np.int32)
        solver.set_constrained_kinematics(kin)
        oracle_ffi.solve(export, 1 / 60, sd, cb)
        ms.absorb(export)
        solver.solve(1 / 60, sd, cb)
        got = ms.to_scene()
        solver.download(got)
        m = pu.compare_scenes(export, got)
        assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], (frame, m)
    assert grew_past_17 and emptied, (grew_past_17, emptied)
    # a lane that is not the reference's to give is refused
    t = next(t for t, tb in ms.batches[threshold].items() if len(tb["refs"]) > 0)
    occupied = next(i for i, lane in enumerate(ms.batches[threshold][t]["refs"]) if lane[0] != -1)
    with pytest.raises(ValueError):  # BEPUHIP_E_INVALID_ARGUMENT
        solver.add_constraint_at(threshold, t, occupied, ms.batches[threshold][t]["refs"][occupied], ms.batches[threshold][t]["prestep"][occupied])
    with pytest.raises(ValueError):  # BEPUHIP_E_INVALID_ARGUMENT
        solver.add_constraint_at(threshold, t, len(ms.batches[threshold][t]["refs"]) + 3 * ms.w, ms.batches[threshold][t]["refs"][occupied], ms.batches[threshold][t]["prestep"][occupied])


def _churn(ms, rng, pair, solver, removals, additions, keep_degree=True):
    """`removals` random contact removals and `additions` random contact additions, on the mirror and on the device."""
    is_contact = lambda t: t in CONTACT_TYPES  # noqa: E731
    for _ in range(removals):
        locs = ms.locations(is_contact)
        bi, t, i = locs[int(rng.integers(len(locs)))]
        if keep_degree and any(not (r & 0x40000000) and ms.dynamic_degree(r) < 2 for r in ms.batches[bi][t]["refs"][i]):
            continue  # (a body that loses its last constraint leaves any island plan)
        ms.remove(bi, t, i)
        solver.remove_constraint(bi, t, i)
    opened = False  # an addition that opens a batch or a type batch: what no island plan absorbs
    for _ in range(additions):
        a, b = pair()
        t = CONTACT_TYPES[int(rng.integers(len(CONTACT_TYPES)))]
        lane = small_scenes.prestep_for(rng, t, ms.bodies[a, 4:7], ms.bodies[b, 4:7])
        held = {(bi, tt) for bi, tbs in enumerate(ms.batches) for tt in tbs}
        bi, index, encoded = ms.add(t, [a, b], lane)
        opened |= (bi, t) not in held
        assert solver.add_constraint(bi, t, encoded, lane) == index
    return opened


def _frame(ms, solver, sd, cb, threads=1):
    export = ms.to_scene()
    oracle_ffi.solve(export, 1 / 60, sd, cb, threads=threads)
    ms.absorb(export)
    solver.solve(1 / 60, sd, cb)
    got = ms.to_scene()
    solver.download(got)
    m = pu.compare_scenes(export, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


@pytest.mark.parametrize("churn_in_flight", [0, 25])
def test_background_replan_with_structural_updates_in_flight_on_a_split_plan(hip_solver_factory, monkeypatch, churn_in_flight):
    """bepuhip_replan_begin / _commit (round 6, VERDICT r5 next #6a): the planner runs on a host thread of the library while the frames go on — here four frames on the
    launch-per-batch rows, each with `churn_in_flight` removals and additions of random contact pairs (most of them across clusters) — and the commit adopts the plan made
    for the constraints of the snapshot, replays the logged operations onto it through the public entry points (reserved slots: the plan absorbs them and the context is on
    the split-island schedule afterwards) and moves the CURRENT prestep data and accumulated impulses — what the frames in between produced — into the new layout. Every
    frame before, between and after is bit-exact against the oracle solving the host mirror; read-backs in the caller's order see the same values through the new layout;
    reset_state returns to uploaded values + structural updates, as it does after bepuhip_replan."""
    monkeypatch.setenv("BEPUHIP_SPLIT_CLUSTERS", "12")
    ms, rng, pair = _build(43, bodies=2600, joints=3000, contacts=5000)
    sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    assert solver.schedule() == 2 and solver.replan_state() == 0
    _frame(ms, solver, sd, cb, 4)
    solver.replan_begin()
    assert solver.schedule() == 0, "from begin to commit the context runs the rows in the caller's order"
    with pytest.raises(native_error()) as e:
        solver.replan_begin()
    assert e.value.code == native_codes().BEPUHIP_E_STATE
    opened = False
    for _ in range(4):
        opened |= _churn(ms, rng, pair, solver, churn_in_flight, churn_in_flight)
        _frame(ms, solver, sd, cb, 4)
    assert solver.replan_state() in (1, 2)
    assert solver.replan_commit(wait=True)
    assert solver.replan_state() == 0
    assert not opened, "the seed was chosen so that no addition of the frames in between opens a type batch (no plan absorbs that: the replay would end on the rows)"
    assert solver.schedule() == 2, "the new plan (with its reserves) took the operations of the frames in between"
    for bi, tbs in enumerate(ms.to_scene().batches):
        for tb in tbs:
            assert solver.constraint_count(bi, tb.type_id) == tb.count
    before = ms.to_scene()
    solver.download(before)
    m = pu.compare_scenes(ms.to_scene(), before)
    assert m["impulses_bit_exact"] and m["prestep_bit_exact"], m
    opened = False
    for _ in range(3):
        opened |= _churn(ms, rng, pair, solver, 10, 10)
        _frame(ms, solver, sd, cb, 4)
    assert opened or (solver.schedule() == 2 and solver.cluster_cycles().size > 1)
    with pytest.raises(native_error()) as e:
        solver.replan_commit(wait=True)  # nothing in flight
    assert e.value.code == native_codes().BEPUHIP_E_STATE


def native_error():
    from bepuphysics2_amd.native import BepuHipError
    return BepuHipError


def native_codes():
    from bepuphysics2_amd import native
    return native


def test_background_replan_on_a_whole_island_plan_and_what_reset_state_returns_to(hip_solver_factory):
    """The hub scene of test_replan_puts_a_context_that_left_its_plan_back_on_the_island_schedule, re-planned in the background: between begin and commit the hub gets more
    constraints (new batches, rows that grow), some are removed again, two frames are solved. After the commit the context is on the whole-island schedule, the frames stay
    bit-exact, and reset_state returns to the uploaded values plus every structural update — the snapshot moved into the new layout with ITS values."""
    rng = np.random.default_rng(6)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-3, 3, 3)) for _ in range(140)]
    ms, pristine = MutableSolver(np.stack(rows)), MutableSolver(np.stack(rows))
    for k in range(1, 139, 2):
        lane = small_scenes.prestep_for(rng, 7, ms.bodies[k, 4:7], ms.bodies[k + 1, 4:7])
        ms.add(7, [k, k + 1], lane); pristine.add(7, [k, k + 1], lane)
    sd, cb = SolveDescription(2, 2), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    assert solver.schedule() == 1

    def hub(k):
        t = [7, 22, 4][k % 3]
        lane = small_scenes.prestep_for(rng, t, ms.bodies[0, 4:7], ms.bodies[k, 4:7])
        bi, index, encoded = ms.add(t, [0, k], lane)
        pristine.add(t, [0, k], lane)
        assert solver.add_constraint(bi, t, encoded, lane) == index

    _frame(ms, solver, sd, cb)
    for k in range(3, 30):
        hub(k)
    _frame(ms, solver, sd, cb)
    assert solver.schedule() == 0
    solver.replan_begin()
    for k in range(30, 36):  # the log: additions that open batches the snapshot did not have ...
        hub(k)
    _frame(ms, solver, sd, cb)
    for bi, t, i in sorted(ms.locations(lambda t: t == 22), reverse=True)[:4]:  # ... and removals (swap-with-last) of constraints the snapshot did have
        ms.remove(bi, t, i); pristine.remove(bi, t, i)
        solver.remove_constraint(bi, t, i)
    _frame(ms, solver, sd, cb)
    assert solver.replan_commit(wait=True)
    # (the replayed additions open batches and type batches: no plan absorbs that, the context is back on the rows — with everything intact)
    for bi, tbs in enumerate(ms.to_scene().batches):
        for tb in tbs:
            assert solver.constraint_count(bi, tb.type_id) == tb.count
    _frame(ms, solver, sd, cb)
    solver.replan_begin()  # nothing happens in flight this time: the commit is bepuhip_replan with the planning elsewhere
    _frame(ms, solver, sd, cb)
    assert solver.replan_commit(wait=True)
    assert solver.schedule() == 1
    _frame(ms, solver, sd, cb)
    _frame(ms, solver, sd, cb)
    assert solver.cluster_cycles().size > 0
    solver.reset_state()
    solver.solve(1 / 60, sd, cb)
    want = pristine.to_scene()
    oracle_ffi.solve(want, 1 / 60, sd, cb)
    got = pristine.to_scene()
    solver.download(got)
    m = pu.compare_scenes(want, got)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"], m


def test_background_replan_is_dropped_by_an_upload_a_replan_or_the_end_of_the_context(hip_solver_factory):
    """A job describes the constraints it was begun for: bepuhip_begin_constraints, bepuhip_replan and bepuhip_destroy drop it (waiting for its worker), bepuhip_replan_cancel
    does so on request; a commit without a job is STATE. Polling without waiting never blocks and reports 0 / 1 / 2."""
    ms, rng, pair = _build(43, bodies=400, joints=500, contacts=900)
    sd, cb = SolveDescription(1, 2), PoseIntegratorCallbacks()
    solver = hip_solver_factory(reserve_update_slots=True)
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    _frame(ms, solver, sd, cb)
    solver.replan_begin()
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)  # drops the job
    assert solver.replan_state() == 0
    _frame(ms, solver, sd, cb)
    solver.replan_begin()
    solver.replan()  # the synchronous call drops it too
    assert solver.replan_state() == 0
    _frame(ms, solver, sd, cb)
    solver.replan_begin()
    solver.replan_cancel()
    assert solver.replan_state() == 0
    _frame(ms, solver, sd, cb)
    solver.replan_begin()
    polls = 0
    while not solver.replan_commit(wait=False):  # frames go on until the worker is done
        _churn(ms, rng, pair, solver, 3, 3)
        _frame(ms, solver, sd, cb)
        polls += 1
        assert polls < 2000
    assert solver.schedule() >= 1
    _frame(ms, solver, sd, cb)
    solver.replan_begin()  # ... and a context that is destroyed with a job in flight waits for the worker and frees it (hip_solver_factory closes the solver)
