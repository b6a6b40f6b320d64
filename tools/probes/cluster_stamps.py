"""Where a step's launch spends its time at the scale of whole clusters (a developer probe; needs a library whose hot_1024 unit was built with -DBEPU_STAMP_WALL=1: every
cluster then leaves the 100 MHz wall clock at its start (low word) and at its end (high word) in the cycles array). BEPUHIP_LIB=<variant> python tools/probes/cluster_stamps.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks

sim = HostSimulation.scene("ragdoll_tube", int(os.environ.get("RAGDOLLS", "15000")), 1, 0, 5)
scene, sd = sim.export(), sim.solve_description()
sim.close()
cb = PoseIntegratorCallbacks()
s = HipSolver()
s.upload(scene)
import time
for _ in range(300):
    s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
t0 = time.perf_counter()
for _ in range(400):
    s.solve(1 / 60, sd, cb, asynchronous=True)
s.sync()
print(f"{scene.constraint_count} constraints: {(time.perf_counter() - t0) / 400 * 1e6:.1f} us per step, back to back")
rows = []
for rep in range(5):
    for _ in range(20):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    raw = s.cluster_cycles().astype(np.uint64)
    start = (raw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    end = (raw >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF
    t0 = start.min()
    st, en = (start - t0) / 100.0, (end - t0) / 100.0  # microseconds
    order = np.argsort(en)
    print(f"launch {rep}: clusters start over {st.max():.2f} us (median {np.median(st):.2f}); ends: first {en.min():.1f}, median {np.median(en):.1f}, 90 % {np.percentile(en, 90):.1f}, last {en.max():.1f} us; "
          f"own span min / median / max {np.min(en - st):.1f} / {np.median(en - st):.1f} / {np.max(en - st):.1f} us")
    xcd = np.arange(raw.size) % 8
    print("   median end per XCD (workgroup id mod 8):", " ".join(f"{np.median(en[xcd == x]):.1f}" for x in range(8)), "| last five to end (cluster id: end):", " ".join(f"{i}:{en[i]:.1f}" for i in order[-5:]))
s.close()
