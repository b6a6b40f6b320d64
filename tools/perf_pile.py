"""Timing of BASELINE.json configs[1] (100k-box pile, one connected island -> launch-per-batch schedule) on the GPU box. Not part of the product."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
from bepuphysics2_amd import roofline

boxes = int(os.environ.get("BOXES", "100000"))
sim = HostSimulation.scene("pile", boxes, 0, 0, 5)
scene, sd = sim.export(), sim.solve_description()
cb = PoseIntegratorCallbacks()
its = scene.constraint_count * int((1 + sd.iterations()).sum())
print(f"pile: {scene.body_count} bodies, {scene.constraint_count} constraints, {len(scene.batches)} batches, substeps {sd.substep_count}, iterations {list(sd.iterations())}")
for b, batch in enumerate(scene.batches):
    print(f"  batch {b}: " + ", ".join(f"type {tb.type_id} x{tb.count}" for tb in batch))
s = HipSolver()
s.upload(scene)
for _ in range(100):
    s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
steps = int(os.environ.get("STEPS", "200"))
t0 = time.perf_counter()
for _ in range(steps):
    s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
ms = (time.perf_counter() - t0) / steps * 1e3
ws, sv, inc = roofline.scene_stage_bytes(scene)
step_bytes = sum(ws + int(k) * sv for k in sd.iterations()) + (sd.substep_count - 1) * inc + roofline.FINAL_BYTES_PER_BODY * scene.body_count
print(f"algorithmic bytes/step {step_bytes / 1e6:.1f} MB -> {step_bytes / ms / 1e9:.3f} TB/s ({step_bytes / ms / 1e9 / 8:.2f} of 8 TB/s)")
print(f"{ms:.3f} ms/step  {its / ms / 1e6:.2f} G c-it/s")
