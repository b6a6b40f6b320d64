"""What structural churn costs on a connected scene (GPU box; not part of the product): every frame the last 1 % of every two-body contact type batch is removed and
as many contacts are added between bodies that were NOT partners before (body A of one removed contact with body B of the next: same batch, so the batch invariant
holds) — pairs that straddle the clusters of the split-island plan like the narrow phase's new contacts do. Reports the frame (calls + solve) and the solve alone,
and whether the context is still on the split plan.
    python tools/perf_churn.py pile|crowd"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pile"
args = {"pile": ("pile", 100000, 0, 0, 5), "crowd": ("ragdoll_tube", 15000, 1, 2, 5)}[name]
sim = HostSimulation.scene(*args)
scene, sd = sim.export(), sim.solve_description()
sim.close()
cb = PoseIntegratorCallbacks()
w = scene.bundle_width
for label, env in (("split plan keeps the updates", {}), ("round 2: updates leave the split plan", {"BEPUHIP_NO_SPLIT_SOFT_UPDATES": "1"})):
    for k, v in env.items():
        os.environ[k] = v
    solver = HipSolver(reserve_update_slots=True, exclusive_device=True)
    solver.upload(scene)
    for _ in range(40):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    contact = [(bi, tb) for bi, b in enumerate(scene.batches) for tb in b if TYPE_TABLE[tb.type_id][3].startswith("Contact") and tb.bodies == 2 and tb.count > 200]
    state = []  # per type batch: the lanes currently at its end: (refs, prestep)
    for bi, tb in contact:
        k = max(2, tb.count // 100)
        refs, pre = tb.refs_lanes(w), tb.prestep_lanes(w)
        state.append([bi, tb.type_id, tb.count, [(refs[i].copy(), pre[i].copy()) for i in range(tb.count - k, tb.count)]])
    calls = sum(len(s[3]) for s in state) * 2

    def churn():
        for s in state:
            bi, t, count, lanes = s
            for j in range(len(lanes)):
                solver.remove_constraint(bi, t, count - 1 - j)
            shifted = [(np.asarray([lanes[i][0][0], lanes[(i + 1) % len(lanes)][0][1]], dtype=np.int32), lanes[i][1]) for i in range(len(lanes))]
            for refs, pre in shifted:
                solver.add_constraint(bi, t, refs, pre)
            s[3] = shifted

    frames = 10
    churn(); solver.solve(1 / 60, sd, cb)
    t0 = time.perf_counter()
    for _ in range(frames):
        churn()
        solver.solve(1 / 60, sd, cb)
    frame_ms = 1e3 * (time.perf_counter() - t0) / frames
    t0 = time.perf_counter()
    for _ in range(50):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    solve_ms = 1e3 * (time.perf_counter() - t0) / 50
    finite = bool(np.isfinite(solver.get_bodies(scene.body_count)).all())
    print(f"{name}, {label}: {calls} structural calls per frame + solve {frame_ms:.2f} ms; the solve alone afterwards {solve_ms:.4f} ms; "
          f"clusters {solver.cluster_cycles().size}; finite {finite}", flush=True)
    solver.close()
    for k in env:
        os.environ.pop(k, None)
