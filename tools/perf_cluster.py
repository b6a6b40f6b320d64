"""Timing experiments for the cluster kernel (GPU box): env knobs x configurations. Not part of the product."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks

ragdolls = int(os.environ.get("RAGDOLLS", "15000"))
sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 0, 5)
scene, sd = sim.export(), sim.solve_description()
cb = PoseIntegratorCallbacks()
its = scene.constraint_count * int((1 + sd.iterations()).sum())


def run(label, env, steps=20, use_clusters=True):
    for k in ("BEPUHIP_DEBUG", "BEPUHIP_CLUSTER_BODIES", "BEPUHIP_CLUSTER_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = HipSolver(use_clusters=use_clusters)
    s.upload(scene)
    for _ in range(3):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:50s} {ms:8.3f} ms/step  {its / ms / 1e6:8.2f} G c-it/s", flush=True)
    s.close()


configs = sys.argv[1:] or ["base"]
for cfg in configs:
    if cfg == "base":
        run("clusters default", {})
        run("global path (launch per batch)", {}, use_clusters=False)
    elif cfg == "debug":
        run("debug=1 (no math)", {"BEPUHIP_DEBUG": "1"})
        run("debug=2 (no global constraint loads)", {"BEPUHIP_DEBUG": "2"})
        run("debug=3 (neither)", {"BEPUHIP_DEBUG": "3"})
    elif cfg == "sizes":
        for cap in (256, 480, 700, 960, 1400):
            for thr in (256, 512):
                run(f"cap={cap} threads={thr}", {"BEPUHIP_CLUSTER_BODIES": str(cap), "BEPUHIP_CLUSTER_THREADS": str(thr)})
