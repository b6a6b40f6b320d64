"""Would fewer, fuller work items pay on the 100k-box pile? (GPU box; an experiment, not part of the product.)

The pile's clusters run 5 items per batch (Contact4 x2, Contact2, Contact3, Contact1) at 35 of 64 lanes. Here every two-body manifold of a batch is rewritten as a Contact4
(the missing contacts repeat contact 0 with a depth far below zero, so they never push) into ONE type batch per batch: the same constraints between the same bodies, 4 items per
batch instead of 5, each constraint at the cost of the largest type. If this is not slower than the typed scene, a merged manifold item with a per-lane contact count — which
would do less arithmetic and read fewer rows than this — is worth building. Results are NOT comparable bit for bit (extra contacts change the friction centre); timing only."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, TypeBatchData, to_aosoa


def as_contact4(tb):
    n = tb.type_id - 3  # Contact1..4 two-body = type ids 4..7
    p, a = tb.prestep_lanes(), tb.accumulated_lanes()
    q = np.zeros((tb.count, 26), np.float32)
    for k in range(4):
        src = k if k < n else 0
        q[:, 4 * k:4 * k + 4] = p[:, 4 * src:4 * src + 4]
        if k >= n:
            q[:, 4 * k + 3] = -1e3
    q[:, 16:26] = p[:, 4 * n:4 * n + 10]
    b = np.zeros((tb.count, 7), np.float32)
    b[:, 0:2], b[:, 2:2 + n], b[:, 6] = a[:, 0:2], a[:, 2:2 + n], a[:, 2 + n]
    return tb.refs_lanes(), q, b


def timed(scene, sd, label):
    s = HipSolver()
    s.upload(scene)
    cb = PoseIntegratorCallbacks()
    for _ in range(100):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    steps = 200
    t0 = time.perf_counter()
    for _ in range(steps):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    print(f"{label}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step ({s.cluster_cycles().size} clusters)", flush=True)
    s.close()


sim = HostSimulation.scene("pile", 100000, 0, 0, 5)
scene, sd = sim.export(), sim.solve_description()
timed(scene, sd, "typed pile (Contact1-4 in their own type batches)")
merged = scene.copy()
for b, batch in enumerate(merged.batches):
    two_body = [tb for tb in batch if 4 <= tb.type_id <= 7]
    rest = [tb for tb in batch if not 4 <= tb.type_id <= 7]
    parts = [as_contact4(tb) for tb in two_body]
    refs, pre, acc = (np.concatenate([p[i] for p in parts]) for i in range(3))
    rest.append(TypeBatchData(7, refs.shape[0], to_aosoa(refs.astype(np.int32), fill=-1), to_aosoa(pre), to_aosoa(acc)))
    merged.batches[b] = rest
timed(merged, sd, "every two-body manifold as a Contact4, one type batch per batch")
