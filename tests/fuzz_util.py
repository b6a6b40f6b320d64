"""The random-scene generators shared by the device fuzzers (tools/fuzz_device.py, tools/fuzz_structural.py, tools/replay_fuzz_device.py, tools/race_hunt.py) and by the
deterministic slices of them that run in the GPU suite (tests/test_gpu_schedule_fuzz.py). Test infrastructure.

Every scene's parameters come from ONE seeded generator that never looks at the device, so (seed, ordinal) names a scene for good: a mismatch the fuzzer reports can be
replayed by ordinal. BEPUHIP_DEBUG_JITTER (a library debugging switch: pseudo-random naps around every work item's wait and publish) turns the island kernels' schedule
inside out without changing what they must compute — results have to stay bit-identical to the oracle's."""
from __future__ import annotations

import json
import os

import numpy as np

import oracle_ffi
import parity_util as pu
import small_scenes
from small_scenes import TYPE_TABLE
from bepuphysics2_amd import native
from bepuphysics2_amd.native import HipSolver, UnsupportedError
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

ALL_TYPES = sorted(TYPE_TABLE.keys())


# ---- random constraint graphs through both schedules (tools/fuzz_device.py) ----
def device_scene_parameters(seed: int, count: int):
    """Random constraint graphs over random subsets of the 44 type ids (sizes from a handful of constraints to split-island plans), kinematic fractions, substep counts with
    uneven iteration schedules, integrator options, angular modes, both schedules, forced split plans, with and without hipGraph."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        p = {"seed": int(rng.integers(1 << 30)), "big": bool(rng.random() < 0.15)}
        p["types"] = [int(t) for t in rng.choice(ALL_TYPES, size=int(rng.integers(1, 10)), replace=False)]
        p["nb"], p["nc"] = (int(rng.integers(3000, 7000)), int(rng.integers(6000, 16000))) if p["big"] else (int(rng.integers(20, 600)), int(rng.integers(10, 2500)))
        p["kin"] = float(rng.choice([0, 0.05, 0.3]))
        p["sub"] = int(rng.integers(1, 6))
        p["its"] = [int(x) for x in rng.integers(1, 4, size=p["sub"])]
        p["cb"] = PoseIntegratorCallbacks(gravity=tuple(rng.uniform(-10, 10, 3)), linear_damping=float(rng.uniform(0, 0.5)), angular_damping=float(rng.uniform(0, 0.5)),
                                          integrate_velocity_for_kinematics=bool(rng.integers(2)), allow_substeps_for_unconstrained_bodies=bool(rng.integers(2)),
                                          angular_integration_mode=int(rng.integers(3)))
        p["use_clusters"] = bool(rng.random() < 0.8)
        p["split_clusters"] = int(rng.integers(16, 40)) if p["big"] else None
        p["frames"] = int(rng.integers(1, 4))
        p["use_graph"] = bool(rng.integers(2))
        out.append(p)
    return out


def describe(p) -> str:
    d = {k: v for k, v in p.items() if k != "cb"}
    cb = p["cb"]
    d["cb"] = {"gravity": [float(g) for g in cb.gravity], "linear_damping": cb.linear_damping, "angular_damping": cb.angular_damping,
               "integrate_velocity_for_kinematics": cb.integrate_velocity_for_kinematics, "allow_substeps_for_unconstrained_bodies": cb.allow_substeps_for_unconstrained_bodies,
               "angular_integration_mode": cb.angular_integration_mode}
    return json.dumps(d)


class environment:
    """Sets developer switches of the library for one run (they are read at upload and at every solve)."""

    def __init__(self, **values):
        self.values = {k: (None if v is None else str(v)) for k, v in values.items()}

    def __enter__(self):
        self.saved = {k: os.environ.get(k) for k in self.values}
        for k, v in self.values.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def build_device_scene(p):
    scene = small_scenes.random_graph_scene(p["seed"], p["nb"], p["nc"], p["types"], kinematic_fraction=p["kin"])
    its = p["its"]
    return scene, SolveDescription(1, p["sub"], velocity_iteration_scheduler=lambda s, its=its: its[s])


def run_device(p, scene, sd, env=None, jitter: int = 0):
    """The scene through libbepuhip.so: (result scene, (schedule, launch policy, clusters of the last solve))."""
    switches = dict(env or {})
    switches["BEPUHIP_SPLIT_CLUSTERS"] = p["split_clusters"]
    switches["BEPUHIP_DEBUG_JITTER"] = jitter or None
    with environment(**switches):
        solver = HipSolver(use_clusters=p["use_clusters"], use_graph=p["use_graph"])
        try:
            if os.environ.get("FUZZ_SPECIALISE") == "1":  # the scene on the unit compiled for exactly its types (bepuhip_specialise_units, waited for: seconds to a minute per new type set)
                got = scene.copy()
                solver.upload(got, sd.fallback_batch_threshold)
                solver.specialise_units(wait=True)
                for _ in range(p["frames"]):
                    solver.solve(1 / 60, sd, p["cb"])
                solver.download(got)
                info = (solver.schedule(), solver.row_policy(), int(solver.cluster_cycles().size), solver.kernel_family())
            else:
                got = pu.run_hip(solver, scene, 1 / 60, sd, p["cb"], frames=p["frames"])
                info = (solver.schedule(), solver.row_policy(), int(solver.cluster_cycles().size))
        finally:
            solver.close()
    return got, info


def exact(ref, got) -> bool:
    m = pu.compare_scenes(ref, got)
    return m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"]


def oracle_is_finite(ref) -> bool:
    """False when the ORACLE's simulation diverged (random stiff constraints, gyroscopic mode): how NaN and infinity spread from there is the hardware's business (payloads,
    min / max of a NaN), not the solver's — such a scene is not compared. Bodies AND constraint words (accumulated impulses can overflow first)."""
    if not np.isfinite(ref.bodies[:, :15]).all():
        return False
    w = ref.bundle_width
    for batch in ref.batches:
        for tb in batch:
            occupied = tb.occupied(w)
            if not (np.isfinite(tb.accumulated_lanes(w)[occupied]).all() and np.isfinite(tb.prestep_lanes(w)[occupied]).all()):
                return False
    return True


def check_device_scene(p, jitter: int = 0):
    """One scene of the generator, device against oracle: 'match', 'mismatch', or 'diverged' (not compared); plus the device's schedule info."""
    scene, sd = build_device_scene(p)
    ref = pu.run_oracle(scene, 1 / 60, sd, p["cb"], frames=p["frames"], threads=4)
    got, info = run_device(p, scene, sd, jitter=jitter)
    if not oracle_is_finite(ref):
        return "diverged", info
    return ("match" if exact(ref, got) else "mismatch"), info


# ---- random add / remove / body-removal / re-plan streams (tools/fuzz_structural.py) ----
STRUCTURAL_TYPES = [4, 5, 6, 7, 22, 25, 30, 47, 0, 3, 23, 46]  # two-body manifolds and joints, one-body manifolds


def run_structural_scene(rng, jitter: int = 0, touch_environment: bool = True, oracle_lock=None) -> dict:
    """One random scene of joints and contact manifolds; every frame a random number of removals (swap-with-last) and additions (random pairs, random types — inside an
    island, across islands, into new batches or type batches, onto reserved slots or not), now and then a body removal (Bodies.RemoveAt's move of the last body) or a
    re-plan, with and without BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS, on both schedules; the device follows through the structural entry points and is compared with the
    oracle solving the host mirror, bit for bit, after every frame. Returns the scene's statistics ('ok': no frame differed).
    ``touch_environment`` False (tests/soak_util.py: several of these at once on threads of one process): the developer switches are left alone — setenv beside another
    thread's getenv is a race of its own; ``oracle_lock``: held around the checker's calls."""
    import contextlib
    from mutable_scene import MutableSolver
    stats = {"ok": True, "frames": 0, "refused": 0, "replans": 0, "background_replans": 0, "body_removals": 0, "report": ""}
    big = rng.random() < 0.3  # one island no workgroup holds: the split-island plan (forced cluster counts so that small scenes split too)
    nb = int(rng.integers(1500, 3500)) if big else int(rng.integers(30, 400))
    nc = int(rng.integers(nb * 2, nb * 4)) if big else int(rng.integers(40, min(900, nb * 12)))  # degrees stay mostly under the fallback threshold (additions to the fallback batch are refused by design)
    split = str(int(rng.integers(8, 32))) if big else None
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-6, 6, 3)) if i % 23 else small_scenes.kinematic_body(rng, rng.uniform(-6, 6, 3)) for i in range(nb)]
    # Round 6: one small scene in five has a LOW FallbackBatchThreshold, so that bodies of higher degree push constraints into the sequential fallback batch and the stream
    # of additions and removals reaches it (bepuhip_add_constraint_at, the fallback branch of Remove: hashed probing, emptied bundles overwritten by the last one)
    threshold = int(rng.integers(3, 7)) if (not big and rng.random() < 0.2) else 64
    stats["fallback_threshold"] = threshold
    ms = MutableSolver(np.stack(rows), fallback_batch_threshold=threshold)

    def add_random(solver=None):
        t = STRUCTURAL_TYPES[int(rng.integers(len(STRUCTURAL_TYPES)))]
        one_body = TYPE_TABLE[t][0] == 1
        while True:
            a, b = (int(x) for x in rng.choice(ms.bodies.shape[0], 2, replace=False))
            if one_body and not ms.is_kinematic(a):
                bodies = [a]
                break
            if not one_body and not (ms.is_kinematic(a) and ms.is_kinematic(b)):
                bodies = [a, b]
                break
        lane = small_scenes.prestep_for(rng, t, ms.bodies[bodies[0], 4:7], ms.bodies[bodies[-1], 4:7])
        bi, index, encoded = ms.add(t, bodies, lane)
        if solver is not None:
            if bi == threshold:
                solver.add_constraint_at(bi, t, index, encoded, lane)
            else:
                assert solver.add_constraint(bi, t, encoded, lane) == index

    for _ in range(nc):
        add_random()
    sub = int(rng.integers(1, 5))
    sd, cb = SolveDescription(int(rng.integers(1, 4)), sub, fallback_batch_threshold=threshold), PoseIntegratorCallbacks()
    with (environment(BEPUHIP_SPLIT_CLUSTERS=split, BEPUHIP_DEBUG_JITTER=jitter or None) if touch_environment else contextlib.nullcontext()):
        solver = HipSolver(use_clusters=bool(rng.random() < 0.8), reserve_update_slots=bool(rng.integers(2)))
        try:
            solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
            replanned_at = []
            in_flight = False
            frames = int(rng.integers(3, 12))
            for frame in range(frames):
                try:
                    for _ in range(int(rng.integers(0, 8))):
                        locs = ms.locations()
                        if len(locs) < 10:
                            break
                        bi, t, i = locs[int(rng.integers(len(locs)))]
                        ms.remove(bi, t, i)
                        solver.remove_constraint(bi, t, i)
                    for _ in range(int(rng.integers(0, 8))):
                        add_random(solver)
                    if rng.random() < 0.25 and ms.bodies.shape[0] > 20:  # Bodies.RemoveAt: a body loses its constraints, the last body takes its slot, its references are patched
                        victim = int(rng.integers(ms.bodies.shape[0]))
                        mine = sorted((loc for loc in ms.locations() if any((int(r) & 0x3FFFFFFF) == victim for r in ms.batches[loc[0]][loc[1]]["refs"][loc[2]])), reverse=True)
                        if len(mine) <= 12:
                            for bi, t, i in mine:
                                ms.remove(bi, t, i)
                                solver.remove_constraint(bi, t, i)
                            for bi, t, i, k, encoded in ms.remove_body(victim):
                                solver.update_body_reference(bi, t, i, k, encoded)
                            solver.set_bodies(ms.bodies)
                            stats["body_removals"] += 1
                    event = rng.random()
                    if in_flight:  # round 6: a re-plan in the background (bepuhip_replan_begin): committed some frames later, with or without waiting for the worker — the operations of the frames in between are replayed onto the new plan
                        if frame == frames - 1 or event < 0.5:
                            if solver.replan_commit(wait=bool(frame == frames - 1 or rng.integers(2))):
                                in_flight = False
                                stats["background_replans"] += 1
                                replanned_at.append(-frame)
                    elif event < 0.15:  # now and then a fresh plan for what the device holds (bepuhip_replan), whatever schedule the context is on
                        solver.replan()
                        stats["replans"] += 1
                        replanned_at.append(frame)
                    elif event < 0.30 and frame < frames - 1:
                        solver.replan_begin()
                        in_flight = True
                    export = ms.to_scene()
                    kin = np.ascontiguousarray(export.constrained_kinematic_indices(), dtype=np.int32)  # Solver.ConstrainedKinematicHandles changes with the constraints: the caller re-sends it
                    native._check(solver.lib, solver.lib.bepuhip_set_constrained_kinematics(solver.ctx, native._ptr(kin), kin.size))
                    with (oracle_lock or contextlib.nullcontext()):
                        oracle_ffi.solve(export, 1 / 60, sd, cb)
                    ms.absorb(export)
                    solver.solve(1 / 60, sd, cb)
                    got = ms.to_scene()
                    solver.download(got)
                    m = pu.compare_scenes(export, got)
                    stats["frames"] += 1
                    if not (m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"]):
                        stats["ok"] = False
                        cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
                        rows = np.flatnonzero((export.bodies[:, cols].view(np.int32) != got.bodies[:, cols].view(np.int32)).any(axis=1))
                        detail = []  # which rows differ: (batch, type id, count, lanes whose impulses / prestep differ, the first such lane's values on either side)
                        for bi, (br, bg) in enumerate(zip(export.batches, got.batches)):
                            for tr, tg in zip(br, bg):
                                occupied = tr.occupied(export.bundle_width)
                                for what, ar, ag in (("impulses", tr.accumulated_lanes(export.bundle_width)[occupied], tg.accumulated_lanes(got.bundle_width)[occupied]),
                                                     ("prestep", tr.prestep_lanes(export.bundle_width)[occupied], tg.prestep_lanes(got.bundle_width)[occupied])):
                                    lanes = np.flatnonzero((np.ascontiguousarray(ar).view(np.int32) != np.ascontiguousarray(ag).view(np.int32)).any(axis=1)) if ar.size else np.zeros(0, np.int64)
                                    if lanes.size and len(detail) < 4:
                                        k = int(lanes[0])
                                        detail.append(f"batch {bi} type {tr.type_id} count {tr.count} {what}: {lanes.size} lanes (first {lanes[:6].tolist()}), lane {k} oracle {np.round(ar[k], 5).tolist()} device {np.round(ag[k], 5).tolist()} "
                                                      f"bodies {tr.refs_lanes(export.bundle_width)[occupied][k].tolist()}")
                        stats["report"] = (" | ".join(detail) + " || " if detail else "") + (f"MISMATCH bodies {nb} constraints {nc} substeps {sub} frame {frame} {m} bodies {rows[:8]} kinematic {[ms.is_kinematic(int(r)) for r in rows[:8]]} "
                                           f"in the caller's constrained-kinematic list {[int(r) in set(kin.tolist()) for r in rows[:8]]} schedule {solver.schedule()} re-planned before frames {replanned_at} (negative: background commit)")
                        break
                except UnsupportedError:  # an addition that lands in the sequential fallback batch: refused by design, the scene ends here
                    stats["refused"] += 1
                    break
            clusters = int(solver.cluster_cycles().size)
        finally:
            solver.close()
    stats["big"] = bool(big)
    stats["on_island_schedule"] = clusters > 0
    stats["on_split_plan"] = bool(big and clusters > 1)
    return stats
