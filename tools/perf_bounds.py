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

# every shape type (2/7 compounds of 1-9 children, 1/7 meshes of 1-59 triangles, 1/3 of the rest hulls of 4-40 points): the lane of a compound or mesh body walks its table entries
from test_bounds import _every_shape_collidables, _random_compounds, _random_hulls, _random_meshes
hulls, meshes = _random_hulls(rng, 64), _random_meshes(rng, 32)
compounds = _random_compounds(rng, 128, len(hulls))
coll = np.tile(_every_shape_collidables(rng, 4200, len(hulls), len(compounds), len(meshes)), n // 4200 + 1)[:n]
s.set_convex_hulls(hulls)
s.set_compounds(compounds)
s.set_meshes(meshes)
s.set_collidables(coll)
for _ in range(5):
    s.predict_bounding_boxes(1 / 60, cb)
t0 = time.perf_counter()
for _ in range(reps):
    s.predict_bounding_boxes(1 / 60, cb)
ms = (time.perf_counter() - t0) / reps * 1e3
print(f"{n} bodies of all nine shape types, resident collidables and tables: {ms:.3f} ms per call end to end")


def timed(label):
    for _ in range(3):
        s.predict_bounding_boxes(1 / 60, cb)
    t0 = time.perf_counter()
    for _ in range(reps):
        s.predict_bounding_boxes(1 / 60, cb)
    print(f"{label}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call end to end")


# when does a body deserve a whole wave? (BEPUHIP_BOUNDS_WAVE_THRESHOLD = entries above which predict_heavy_bounds_kernel takes the body; default 64)
for threshold in (8, 16, 32, 64, 1 << 30):
    os.environ["BEPUHIP_BOUNDS_WAVE_THRESHOLD"] = str(threshold)
    timed(f"  same bodies, wave threshold {threshold}")
# a few bodies with very large shapes among many small ones: a 5,000-child compound, a 20,000-triangle mesh, a 2,000-point hull, 50 bodies each
from bepuphysics2_amd.native import COMPOUND_CHILD_DTYPE
big = np.zeros(5000, dtype=COMPOUND_CHILD_DTYPE)
big["shape_type"], big["shape"][:, :3], big["local_position"], big["local_orientation"][:, 3] = 2, 0.3, rng.uniform(-20, 20, (5000, 3)), 1
s.set_compounds(compounds + [big])
s.set_meshes(meshes + [(rng.normal(size=(20000, 3, 3)).astype(np.float32), np.ones(3, np.float32))])
s.set_convex_hulls(hulls + [rng.normal(size=(2000, 3)).astype(np.float32)])
for k in range(50):
    for j, (t, table) in enumerate(((6, len(compounds)), (8, len(meshes)), (5, len(hulls)))):
        i = 1000 + 4000 * k + 97 * j
        coll["shape_type"][i], coll["shape"][i, 0] = t, table
s.set_collidables(coll)
for threshold in (64, 1 << 30):
    os.environ["BEPUHIP_BOUNDS_WAVE_THRESHOLD"] = str(threshold)
    timed(f"  plus 150 bodies with very large shapes, wave threshold {threshold}")
