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


def run(label, env, steps=int(os.environ.get('STEPS', '400')), use_clusters=True, use_graph=True):
    for k in ("BEPUHIP_DEBUG", "BEPUHIP_CLUSTER_BODIES", "BEPUHIP_CLUSTER_THREADS", "BEPUHIP_CLUSTER_VARIANT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = HipSolver(use_clusters=use_clusters, use_graph=use_graph)
    s.upload(scene)
    for _ in range(int(os.environ.get('WARM', '200'))):  # long warm-up: the clock needs tens of milliseconds of load to ramp up
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    cyc = s.cluster_cycles()
    extra = f"  cluster kcycles min/mean/max {cyc.min() / 1e3:.0f}/{cyc.mean() / 1e3:.0f}/{cyc.max() / 1e3:.0f} of {cyc.size} => {cyc.max() / ms / 1e6:.2f} GHz" if cyc.size else ""
    print(f"{label:34s} {ms:8.3f} ms/step  {its / ms / 1e6:8.2f} G c-it/s{extra}", flush=True)
    s.close()


configs = sys.argv[1:] or ["base"]
for cfg in configs:
    if cfg == "clusters":
        run("clusters default", {})
    elif cfg == "graph":
        for _ in range(2):
            run("clusters, hipGraph replay", {})
            run("clusters, direct launch", {}, use_graph=False)
    elif cfg == "base":
        run("clusters default", {})
        run("global path (launch per batch)", {}, use_clusters=False)
    elif cfg == "waves":
        for thr in (512, 768, 1024, 512, 768, 1024):
            run(f"threads={thr}", {"BEPUHIP_CLUSTER_THREADS": str(thr)})
    elif cfg == "coresident":
        # Several smaller workgroups per CU instead of one: the register budget is the 1024-thread build's (128 VGPRs) or the 512-thread build's (170),
        # the LDS follows the cluster size. BEPUHIP_CLUSTER_VARIANT is read once per process: one child process per line.
        import subprocess
        per_ragdoll = 16  # bodies of one ragdoll: clusters hold whole ragdolls
        base = (ragdolls * per_ragdoll + 247) // 248
        for div, thr, variant in ((1, 1024, 1024), (2, 512, 1024), (2, 512, 512), (2, 1024, 1024), (3, 384, 1024), (4, 256, 1024), (4, 256, 512), (4, 512, 1024), (8, 128, 1024), (8, 256, 1024)):
            cap = max(per_ragdoll, (base // div + per_ragdoll - 1) // per_ragdoll * per_ragdoll)
            env = dict(os.environ, BEPUHIP_CLUSTER_BODIES=str(cap), BEPUHIP_CLUSTER_THREADS=str(thr), BEPUHIP_CLUSTER_VARIANT=str(variant), LABEL=f"1/{div} cluster ({cap} bodies), {thr} threads, {variant} budget")
            subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env)
    elif cfg == "one":
        run(os.environ["LABEL"], {k: os.environ[k] for k in ("BEPUHIP_CLUSTER_BODIES", "BEPUHIP_CLUSTER_THREADS", "BEPUHIP_CLUSTER_VARIANT")})
    elif cfg == "sizes":
        for cap in (256, 480, 700, 960, 1400):
            for thr in (256, 512):
                run(f"cap={cap} threads={thr}", {"BEPUHIP_CLUSTER_BODIES": str(cap), "BEPUHIP_CLUSTER_THREADS": str(thr)})
