"""GPU box: re-uploads of the headline scene, nothing else — `end_constraints` per call (and its phases with BEPUHIP_PLAN_STATS=1/2) under the environment the caller
sets (BEPUHIP_PLAN_THREADS, taskset ...):  python tools/perf_upload.py [ragdolls] [repeats]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

import bench
from bepuphysics2_amd.native import HipSolver, _check, _ptr

ragdolls = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 7
scene, sd = bench.build_scene(ragdolls, 5)
solver = HipSolver(device=0)
work = scene.copy()
solver.register_host_memory(work.bodies)
for b in work.batches:
    for tb in b:
        if tb.count:
            solver.register_host_memory(tb.prestep)
            solver.register_host_memory(tb.accumulated)
solver.upload(work)
kin = np.ascontiguousarray(scene.constrained_kinematic_indices(), dtype=np.int32)
ms, parts = [], []
for _ in range(repeats):
    solver.set_bodies(work.bodies)
    t1 = time.perf_counter()
    _check(solver.lib, solver.lib.bepuhip_begin_constraints(solver.ctx, len(work.batches), sd.fallback_batch_threshold))
    t2 = time.perf_counter()
    for bi, batch in enumerate(work.batches):
        for tb in batch:
            _check(solver.lib, solver.lib.bepuhip_set_type_batch(solver.ctx, bi, tb.type_id, tb.count, _ptr(tb.body_refs), _ptr(tb.prestep), _ptr(tb.accumulated)))
    t3 = time.perf_counter()
    _check(solver.lib, solver.lib.bepuhip_end_constraints(solver.ctx))
    t4 = time.perf_counter()
    _check(solver.lib, solver.lib.bepuhip_set_constrained_kinematics(solver.ctx, _ptr(kin), kin.size))
    t5 = time.perf_counter()
    ms.append(1e3 * (t5 - t1))
    parts.append([1e3 * (b - a) for a, b in ((t1, t2), (t2, t3), (t3, t4), (t4, t5))])
print("end_constraints_ms:", " ".join(f"{m:.2f}" for m in ms), "| median", f"{sorted(ms)[len(ms) // 2]:.2f}", "| threads", os.environ.get("BEPUHIP_PLAN_THREADS", "default"),
      "| cpus", len(os.sched_getaffinity(0)))
print("begin / set_type_batch (all) / end / set_constrained_kinematics, medians:", " ".join(f"{sorted(p[k] for p in parts)[len(parts) // 2]:.2f}" for k in range(4)))
solver.close()
