"""Timing of bepuhip_predict_bounding_boxes on the bench scene's body count (GPU box). Not part of the product."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from test_bounds import _random_bodies, _random_collidables
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks

n = int(os.environ.get("BODIES", "240000"))
rng = np.random.default_rng(1)
bodies = np.tile(_random_bodies(rng, 4000), (n // 4000 + 1, 1))[:n]
coll = np.tile(_random_collidables(rng, 4000), n // 4000 + 1)[:n]
s = HipSolver()
s.set_bodies(bodies)
cb = PoseIntegratorCallbacks()
for _ in range(5):
    s.predict_bounding_boxes(1 / 60, cb, coll)
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    s.predict_bounding_boxes(1 / 60, cb, coll)
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"{n} bodies: {ms:.3f} ms per call end to end (64 B/body up, 32 B/body down over PCIe included)")
s.set_collidables(coll)
for _ in range(5):
    s.predict_bounding_boxes(1 / 60, cb)
t0 = time.perf_counter()
for _ in range(reps):
    s.predict_bounding_boxes(1 / 60, cb)
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"{n} bodies, resident collidables: {ms:.3f} ms per call end to end (32 B/body down)")
