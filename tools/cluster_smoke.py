"""Small cluster-schedule parity probes against the oracle (GPU box), each bounded: kernel-debugging aid, not part of the product."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import parity_util as pu
import small_scenes
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

cb = PoseIntegratorCallbacks()
for name, scene, sd in (
    ("ballsocket graph 300/700", small_scenes.random_graph_scene(122, 300, 700, [22]), SolveDescription(2, 8)),
    ("contact4 graph 300/700", small_scenes.random_graph_scene(107, 300, 700, [7]), SolveDescription(2, 8)),
    ("mixed graph 200/500", small_scenes.random_graph_scene(7, 200, 500, [0, 7, 22, 25, 30, 47]), SolveDescription(1, 4)),
):
    t0 = time.time()
    solver = HipSolver()
    try:
        ref = pu.run_oracle(scene, 1 / 60, sd, cb)
        got = pu.run_hip(solver, scene, 1 / 60, sd, cb)
        m = pu.compare_scenes(ref, got)
        print(name, "batches", len(scene.batches), {k: m[k] for k in ("bodies_bit_exact", "impulses_bit_exact", "prestep_bit_exact", "bodies_max_ulp")}, f"{time.time() - t0:.1f}s", flush=True)
    except Exception as e:  # noqa: BLE001
        print(name, "FAILED:", e, f"{time.time() - t0:.1f}s", flush=True)
    solver.close()
