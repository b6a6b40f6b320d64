"""What a different batch colouring buys the island schedule (GPU box): RagdollTubeBenchmark as the host builds it vs recoloured largest-degree-first on the device."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd import colouring
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks

sim = HostSimulation.scene("ragdoll_tube", int(os.environ.get("RAGDOLLS", "15000")), 1, int(os.environ.get("CONTACTS", "0")), 5)
scene, sd = sim.export(), sim.solve_description()
cb = PoseIntegratorCallbacks()
_types, flat, _pre, _acc = colouring.flatten_constraints(scene)
for order in (0, 1):
    colouring.colour_constraints(flat, scene.body_count, order)  # untimed: device and library initialisation
    t0 = time.perf_counter()
    _colours, nbatches, nrounds = colouring.colour_constraints(flat, scene.body_count, order)
    print(f"bepuhip_colour_constraints order {order}: {flat.shape[0]} constraints -> {nbatches} batches in {nrounds} rounds, {1e3 * (time.perf_counter() - t0):.1f} ms (native call, H2D + D2H included)")
t0 = time.perf_counter()
recoloured, rounds = colouring.recolour_scene(scene)
print(f"max dynamic degree {colouring.max_dynamic_degree(scene)}; host first fit {len(scene.batches)} batches; device largest-degree-first {len(recoloured.batches)} batches "
      f"in {rounds} rounds ({time.perf_counter() - t0:.2f} s including the host-side regrouping)")
for name, sc in (("host colouring", scene), ("recoloured", recoloured)):
    s = HipSolver()
    s.upload(sc)
    for _ in range(100):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    cyc = s.cluster_cycles()
    print(f"{name}: {ms:.4f} ms/step, cluster kcycles mean {cyc.mean() / 1e3 if cyc.size else 0:.1f} max {cyc.max() / 1e3 if cyc.size else 0:.1f}")
    s.close()
