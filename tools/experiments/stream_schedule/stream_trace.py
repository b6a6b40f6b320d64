"""Per-hop timeline of the stream schedule on the pile (BASELINE.json configs[1]): where a hop's time goes. Not part of the product."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks

sim = HostSimulation.scene("pile", int(os.environ.get("BOXES", "100000")), 0, 0, 5)
scene, sd = sim.export(), sim.solve_description()
cb = PoseIntegratorCallbacks()
s = HipSolver(use_stream=True)
s.upload(scene)
for _ in range(20):
    s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
s.set_cluster_trace(True)
s.solve(1 / 60, sd, cb)
t = s.stream_trace().astype(np.int64)
hops = int((t[0, :, 0] > 0).sum())
t0 = t[0, 0, 0]
print(f"{hops} hops; stamps in us relative to the first; per traced wavefront: reached / gate passed / blocks done / arrived")
for h in range(min(hops, int(os.environ.get("HOPS", "30")))):
    row = []
    for w in range(3):
        if t[w, h, 0] == 0:
            row.append("      (no block)            ")
            continue
        r = (t[w, h] - t0) / 100.0
        row.append(f"{r[0]:8.2f} +{r[1] - r[0]:6.2f} +{r[2] - r[1]:6.2f} +{r[3] - r[2]:5.2f}")
    print(f"hop {h:3d}: " + " | ".join(row))
g = t[0, :hops]
wait, work, arrive = (g[:, 1] - g[:, 0]) / 100.0, (g[:, 2] - g[:, 1]) / 100.0, (g[:, 3] - g[:, 2]) / 100.0
print(f"wavefront 0 means over {hops} hops: wait {wait.mean():.2f} us, blocks {work.mean():.2f} us, arrive {arrive.mean():.2f} us; whole launch {(g[-1, 3] - g[0, 0]) / 100.0:.1f} us")
