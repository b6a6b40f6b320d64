"""Fuzzer of the HIP path against the scalar oracle (GPU box; a developer tool, not part of the product or of the test suite).

Random constraint graphs over random subsets of the 44 type ids (sizes from a handful of constraints to split-island plans), kinematic fractions, substep counts with uneven
iteration schedules, integrator options, angular modes, both schedules, forced split plans: `frames` frames through oracle/ and through libbepuhip.so, compared bit for bit.
    python tools/fuzz_device.py <seed> <seconds>"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import parity_util as pu
import small_scenes
from small_scenes import TYPE_TABLE
from bepuphysics2_amd.native import HipSolver, UnsupportedError
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

ALL = sorted(TYPE_TABLE.keys())
TWO_BODY = [t for t in ALL if TYPE_TABLE[t][0] <= 2]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
n = bad = split = batch_path = refused = diverged = 0
while time.time() < t_end:
    seed = int(rng.integers(1 << 30))
    big = rng.random() < 0.15  # large enough for a split-island plan (every type id since round 3: three- and four-body constraints are split too)
    pool = ALL
    types = [int(t) for t in rng.choice(pool, size=int(rng.integers(1, 10)), replace=False)]
    nb, nc = (int(rng.integers(3000, 7000)), int(rng.integers(6000, 16000))) if big else (int(rng.integers(20, 600)), int(rng.integers(10, 2500)))
    kin = float(rng.choice([0, 0.05, 0.3]))
    scene = small_scenes.random_graph_scene(seed, nb, nc, types, kinematic_fraction=kin)
    sub = int(rng.integers(1, 6))
    its = [int(x) for x in rng.integers(1, 4, size=sub)]
    sd = SolveDescription(1, sub, velocity_iteration_scheduler=lambda s: its[s])
    cb = PoseIntegratorCallbacks(gravity=tuple(rng.uniform(-10, 10, 3)), linear_damping=float(rng.uniform(0, 0.5)), angular_damping=float(rng.uniform(0, 0.5)),
                                 integrate_velocity_for_kinematics=bool(rng.integers(2)), allow_substeps_for_unconstrained_bodies=bool(rng.integers(2)),
                                 angular_integration_mode=int(rng.integers(3)))
    use_clusters = rng.random() < 0.8
    if big:
        os.environ["BEPUHIP_SPLIT_CLUSTERS"] = str(int(rng.integers(16, 40)))
    else:
        os.environ.pop("BEPUHIP_SPLIT_CLUSTERS", None)
    frames = int(rng.integers(1, 4))
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=frames, threads=4)
    solver = HipSolver(use_clusters=use_clusters, use_graph=bool(rng.integers(2)))
    try:
        got = pu.run_hip(solver, scene, 1 / 60, sd, cb, frames=frames)
    except UnsupportedError:  # round 2 refused a sequential fallback batch together with a momentum-conserving angular mode; nothing should be refused any more
        refused += 1
        solver.close()
        continue
    clusters = solver.cluster_cycles().size
    solver.close()
    split += big and clusters > 1
    batch_path += clusters == 0
    if not np.isfinite(ref.bodies[:, :15]).all():  # the ORACLE's simulation diverged (random stiff constraints, gyroscopic mode): how NaN and infinity spread from there is the
        diverged += 1                              # hardware's business (payloads, min / max of a NaN), not the solver's — nothing to compare
        continue
    m = pu.compare_scenes(ref, got)
    n += 1
    if not (m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"]):
        bad += 1
        print("MISMATCH seed", seed, types, nb, nc, kin, sub, its, use_clusters, frames, cb, m, flush=True)
print(f"scenes {n} (split-island plans {split}, launch-per-batch {batch_path}, refused as UNSUPPORTED {refused}), diverged in the oracle and not compared {diverged}, mismatches {bad}")
