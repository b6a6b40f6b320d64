"""Fuzzer of structural updates on the device against the oracle solving the host mirror (GPU box; a developer tool, not part of the product or of the test suite).

Random scenes of joints and contact manifolds; every frame a random number of removals (swap-with-last) and additions (random pairs, random types — inside an island,
across islands, into new batches or type batches, onto reserved slots or not, whatever comes), with and without BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS, on both schedules; the
device follows through bepuhip_add_constraint / remove_constraint and is compared with the oracle bit for bit after every frame.
    python tools/fuzz_structural.py <seed> <seconds>"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import oracle_ffi
import parity_util as pu
import small_scenes
from bepuphysics2_amd import native
from bepuphysics2_amd.native import HipSolver, UnsupportedError
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription
from mutable_scene import MutableSolver

TYPES = [4, 5, 6, 7, 22, 25, 30, 47, 0, 3, 23, 46]  # two-body manifolds and joints, one-body manifolds
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
scenes = frames_total = bad = stayed = refused = 0
split_scenes = split_stayed = replans = body_removals = 0
while time.time() < t_end:
    big = rng.random() < 0.3  # one island no workgroup holds: the split-island plan (two-body types and one-body manifolds; forced cluster counts so that small scenes split too)
    nb = int(rng.integers(1500, 3500)) if big else int(rng.integers(30, 400))
    nc = int(rng.integers(nb * 2, nb * 4)) if big else int(rng.integers(40, min(900, nb * 12)))  # degrees stay mostly under the fallback threshold (additions to the fallback batch are refused by design)
    if big:
        os.environ["BEPUHIP_SPLIT_CLUSTERS"] = str(int(rng.integers(8, 32)))
    else:
        os.environ.pop("BEPUHIP_SPLIT_CLUSTERS", None)
    rows = [small_scenes.random_dynamic_body(rng, rng.uniform(-6, 6, 3)) if i % 23 else small_scenes.kinematic_body(rng, rng.uniform(-6, 6, 3)) for i in range(nb)]
    ms = MutableSolver(np.stack(rows))

    def add_random(solver=None):
        t = TYPES[int(rng.integers(len(TYPES)))]
        one_body = small_scenes.TYPE_TABLE[t][0] == 1
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
            assert solver.add_constraint(bi, t, encoded, lane) == index

    for _ in range(nc):
        add_random()
    sub = int(rng.integers(1, 5))
    sd, cb = SolveDescription(int(rng.integers(1, 4)), sub), PoseIntegratorCallbacks()
    solver = HipSolver(use_clusters=bool(rng.random() < 0.8), reserve_update_slots=bool(rng.integers(2)))
    solver.upload(ms.to_scene(), sd.fallback_batch_threshold)
    ok = True
    replanned_at = []
    for frame in range(int(rng.integers(3, 12))):
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
                  body_removals += 1
          if rng.random() < 0.15:  # now and then a fresh plan for what the device holds (bepuhip_replan), whatever schedule the context is on
              solver.replan()
              replans += 1
              replanned_at.append(frame)
          export = ms.to_scene()
          kin = np.ascontiguousarray(export.constrained_kinematic_indices(), dtype=np.int32)  # Solver.ConstrainedKinematicHandles changes with the constraints: the caller re-sends it
          native._check(solver.lib, solver.lib.bepuhip_set_constrained_kinematics(solver.ctx, native._ptr(kin), kin.size))
          oracle_ffi.solve(export, 1 / 60, sd, cb)
          ms.absorb(export)
          solver.solve(1 / 60, sd, cb)
          got = ms.to_scene()
          solver.download(got)
          m = pu.compare_scenes(export, got)
          frames_total += 1
          if not (m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"]):
              ok = False
              cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
              rows = np.flatnonzero((export.bodies[:, cols].view(np.int32) != got.bodies[:, cols].view(np.int32)).any(axis=1))
              print("MISMATCH", nb, nc, sub, frame, m, "bodies", rows[:8], "kinematic", [ms.is_kinematic(int(r)) for r in rows[:8]], "in the caller's constrained-kinematic list", [int(r) in set(kin.tolist()) for r in rows[:8]],
                    "schedule", solver.schedule(), "re-planned before frames", replanned_at, flush=True)
              break
      except UnsupportedError:  # an addition that lands in the sequential fallback batch: refused by design, the scene ends here
        refused += 1
        break
    stayed += solver.cluster_cycles().size > 0
    split_scenes += big
    split_stayed += big and solver.cluster_cycles().size > 1
    solver.close()
    scenes += 1
    bad += not ok
print(f"re-plans {replans}, body removals {body_removals}; scenes {scenes} ({split_scenes} big enough for a split-island plan, {split_stayed} of them still on it at the end), frames {frames_total}, still on an island schedule at the end {stayed}, "
      f"ended by a refused fallback-batch addition {refused}, mismatching scenes {bad}")
