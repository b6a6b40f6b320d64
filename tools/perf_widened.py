"""Timings of scenes with the widened constraint types (GPU box; not part of the product): they run the second `cluster_kernel` variant, by default with 1024 threads
per workgroup like the lean one (its 1024-thread build spills 700 VGPRs and still beats the 512-thread build on the pool's slow class of box; both are timed here).
* rigs: the bench scene's 15,000 ragdolls (same bodies, same constraint graph, same batches) with its seven joint types other than BallSocket replaced by widened ones
  (AngularSwivelHinge, DistanceLimit, AngularServo, TwistMotor, AngularAxisMotor, Weld, BallSocketServo; random settings) — type batches as long as the headline's;
* all 44 type ids drawn at random in islands of 16 bodies with 64 constraints: 790 type batches, one or two constraints per cluster and type batch — the worst case
  for work items of up to 64 lanes.
    python tools/perf_widened.py [ragdolls] [islands]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import small_scenes  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks, SolveDescription  # noqa: E402

ragdolls = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
islands = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
from bepuphysics2_amd.synthetic import rig_scene  # noqa: E402

sd = SolveDescription(1, 4)
cb = PoseIntegratorCallbacks()
for label in ("rigs (the bench scene's graph, seven joint types widened)", "all 44 type ids at random"):
    t0 = time.perf_counter()
    scene = rig_scene(ragdolls)[0] if label.startswith("rigs") else small_scenes.island_scene(11, islands, 16, 64, sorted(TYPE_TABLE))
    build_s = time.perf_counter() - t0
    for threads in ("", "512"):
        if threads:
            os.environ["BEPUHIP_CLUSTER_THREADS"] = threads
        solver = HipSolver(exclusive_device=True)
        solver.upload(scene)
        for _ in range(40):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.reset_state(); solver.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 50
        its = sd.iterations()
        per_step = scene.constraint_count * int((1 + its).sum())
        print(f"{label}: {scene.body_count} bodies, {scene.constraint_count} constraints, {len(scene.batches)} batches (scene built in {build_s:.0f} s); "
              f"threads {threads or 'default'}: {ms:.4f} ms/step, {per_step / ms / 1e6:.2f} G constraint-iterations/s, clusters {solver.cluster_cycles().size}, finite {bool(np.isfinite(solver.get_bodies(scene.body_count)).all())}", flush=True)
        solver.close()
        os.environ.pop("BEPUHIP_CLUSTER_THREADS", None)
