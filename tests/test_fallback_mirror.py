"""The host mirror's sequential-fallback-batch rules (tests/mutable_scene.py: TypeProcessor.AllocateInTypeBatchForFallback, TypeProcessor.cs:451-571, and the fallback
branch of TypeProcessor.Remove, :633-694) — CPU only: the invariants the reference's debug validators check (ValidateEmptyFallbackSlots, ValidateFallbackAccessSafety,
:385-430), and that the oracle solves what the mirror exports."""
import numpy as np

import oracle_ffi
import small_scenes
import wide_ffi
from mutable_scene import MutableSolver
from bepuphysics2_amd.scene import KINEMATIC_MASK, TYPE_TABLE, PoseIntegratorCallbacks, SolveDescription

THRESHOLD = 4


def _star(rng, hubs=3, spokes=90):
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-1, 1, 3)) for _ in range(hubs)]
    rows += [small_scenes.random_dynamic_body(rng, rng.uniform(-4, 4, 3)) if i % 9 else small_scenes.kinematic_body(rng, rng.uniform(-4, 4, 3)) for i in range(spokes)]
    return MutableSolver(np.stack(rows), fallback_batch_threshold=THRESHOLD), hubs, spokes


def _add_spoke(ms, rng, hubs, spokes, types=(7, 22, 4, 47, 30)):
    t = types[int(rng.integers(len(types)))]
    hub, spoke = int(rng.integers(hubs)), hubs + int(rng.integers(spokes))
    pair = [hub, spoke] if rng.integers(2) else [spoke, hub]
    return ms.add(t, pair, small_scenes.prestep_for(rng, t, ms.bodies[pair[0], 4:7], ms.bodies[pair[1], 4:7]))


def _check_invariants(ms):
    fb = ms.batches[THRESHOLD] if len(ms.batches) > THRESHOLD else {}
    for t, tb in fb.items():
        refs, handles = tb["refs"], tb["handles"]
        assert len(refs) == len(handles) == len(tb["prestep"]) == len(tb["acc"])
        # (ConstraintCount is recomputed only when a removal empties a bundle, :681-682: the last bundle may end in empty lanes)
        for i, lane in enumerate(refs):
            assert (lane[0] == -1) == (handles[i] == -1), "ValidateEmptyFallbackSlots"
            assert all(r == -1 for r in lane) or all(r >= 0 for r in lane)
        for b0 in range(0, len(refs), ms.w):  # ValidateFallbackAccessSafety: no dynamic body twice inside a bundle
            dynamic = [r for lane in refs[b0:b0 + ms.w] if lane[0] != -1 for r in lane if not (r & KINEMATIC_MASK)]
            assert len(dynamic) == len(set(dynamic)), (t, b0)
            if b0 + ms.w < len(refs):
                assert any(lane[0] != -1 for lane in refs[b0:b0 + ms.w]), "an empty bundle in the middle would have been overwritten by the last one"
        live = [h for h in handles if h != -1]
        assert len(live) == len(set(live))


def test_fallback_allocation_and_removal_keep_the_references_validators_quiet():
    rng = np.random.default_rng(3)
    ms, hubs, spokes = _star(rng)
    for _ in range(260):
        _add_spoke(ms, rng, hubs, spokes)
    assert len(ms.batches) == THRESHOLD + 1 and sum(len(tb["refs"]) for tb in ms.batches[THRESHOLD].values()) > 150
    _check_invariants(ms)
    # more than 17 bundles in at least one type batch: the hashed probing ran
    assert max(len(tb["refs"]) for tb in ms.batches[THRESHOLD].values()) > 17 * ms.w
    for step in range(400):
        locs = [loc for loc in ms.locations() if loc[0] == THRESHOLD]
        if rng.random() < 0.55 and len(locs) > 5:
            ms.remove(*locs[int(rng.integers(len(locs)))])
        else:
            _add_spoke(ms, rng, hubs, spokes)
        _check_invariants(ms)


def test_rehash_matches_the_reference_on_known_values():
    """HashHelper.Rehash (QuickDictionary.cs:20-41): computed by hand from the C# (uint multiply by 982451653, three rotations xor'ed)."""
    assert MutableSolver.rehash(0) == 0
    u = (1 * 982451653) & 0xFFFFFFFF
    rot = lambda x, k: ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF  # noqa: E731
    want = rot(u, 6) ^ rot(u, 13) ^ rot(u, 25)
    assert MutableSolver.rehash(1) == (want - (1 << 32) if want & 0x80000000 else want)
    assert -(1 << 31) <= MutableSolver.rehash(123456789) < (1 << 31)


def test_both_oracles_solve_what_the_mirror_exports_after_churn():
    rng = np.random.default_rng(4)
    ms, hubs, spokes = _star(rng, hubs=2, spokes=60)
    for _ in range(120):
        _add_spoke(ms, rng, hubs, spokes)
    for _ in range(60):
        locs = [loc for loc in ms.locations() if loc[0] == THRESHOLD]
        ms.remove(*locs[int(rng.integers(len(locs)))])
        _add_spoke(ms, rng, hubs, spokes)
    sd, cb = SolveDescription(1, 3, fallback_batch_threshold=THRESHOLD), PoseIntegratorCallbacks()
    a, b = ms.to_scene(), ms.to_scene()
    oracle_ffi.solve(a, 1 / 60, sd, cb)
    wide_ffi.solve(b, 1 / 60, sd, cb)
    assert np.isfinite(a.bodies[:, :15]).all()
    cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]  # (the padding lanes of BodyDynamics are not defined)
    assert np.array_equal(a.bodies[:, cols].view(np.int32), b.bodies[:, cols].view(np.int32))
