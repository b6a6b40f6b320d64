"""The momentum-conserving angular modes on the bench scene (GPU box; not part of the product): the island schedule's conserving kernel units against the launch-per-batch
schedule they replaced (BEPUHIP_CONSERVING_CLUSTERS=0), and the nonconserving mode for scale.
    python tools/perf_conserving.py [ragdolls]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import PoseIntegratorCallbacks  # noqa: E402

sim = HostSimulation.scene("ragdoll_tube", int(sys.argv[1]) if len(sys.argv) > 1 else 15000, 1, 0, 5)
scene, sd = sim.export(), sim.solve_description()
sim.close()
reference = {}
for mode in (0, 1, 2):
    for clusters in ("1", "0") if mode else ("1",):
        os.environ["BEPUHIP_CONSERVING_CLUSTERS"] = clusters
        cb = PoseIntegratorCallbacks(angular_integration_mode=mode)
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
        bodies = solver.get_bodies(scene.body_count)
        same = "" if mode not in reference else f", bit-identical to the other schedule: {bool(np.array_equal(reference[mode].view(np.int32), bodies.view(np.int32)))}"
        reference.setdefault(mode, bodies)
        schedule = "island schedule" if (mode == 0 or clusters == "1") else "launch-per-batch (BEPUHIP_CONSERVING_CLUSTERS=0)"
        print(f"angular mode {mode}, {schedule}: {ms:.4f} ms/step{same}", flush=True)
        solver.close()
