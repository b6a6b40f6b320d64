"""What structural churn costs on a connected scene (GPU box; not part of the product). Every frame a hundredth of every two-body contact type batch is removed (every
hundredth constraint, another residue every frame: the churn is spread over the scene) and as many contacts come back: most between the same bodies (the narrow phase's
refresh of a persisting pair), every NEW_PAIRS-th between the first body and a NEAR one (the nearest body index its batch does not hold yet) — a pair that may straddle
the clusters of the split-island plan, like the narrow phase's new contacts do. tools/plan_harness runs the same churn on the plan's host mirrors without a device and
validates the plan after every frame. Reports the frame (calls + solve) and the solve alone, and whether the context is still on the split plan.
    python tools/perf_churn.py pile|crowd [frames]"""
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
new_pairs = int(os.environ.get("NEW_PAIRS", "5"))
for label, env in (("split plan keeps the updates", {}), ("round 2: updates leave the split plan", {"BEPUHIP_NO_SPLIT_SOFT_UPDATES": "1"})):
    for k, v in env.items():
        os.environ[k] = v
    solver = HipSolver(reserve_update_slots=True, exclusive_device=True)
    solver.upload(scene)
    for _ in range(40):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    contact = [(bi, tb) for bi, b in enumerate(scene.batches) for tb in b if TYPE_TABLE[tb.type_id][3].startswith("Contact") and tb.bodies == 2 and tb.count > 200]
    in_batch = []  # per batch: which bodies it references
    for b in scene.batches:
        held = np.zeros(scene.body_count + 1, dtype=bool)
        for tb in b:
            r = tb.refs_lanes(w)[: tb.count].reshape(-1)
            held[r[(r >= 0) & (r < (1 << 30))]] = True
        in_batch.append(held)
    state = [[bi, tb.type_id, [r.copy() for r in tb.refs_lanes(w)[: tb.count]], [p.copy() for p in tb.prestep_lanes(w)[: tb.count]]] for bi, tb in contact]  # in the caller's order
    calls = 2 * sum(len(s[2]) // 100 for s in state)
    frame_number = [0]

    def churn():  # the frame's calls are listed first: the timed part is the library's, not this generator's
        ops = []
        frame = frame_number[0]
        frame_number[0] += 1
        for bi, t, refs, pre in state:
            held = in_batch[bi]
            n = len(refs) // 100
            stride = len(refs) // n
            first = (frame * 37) % stride
            lanes = []
            for i in range(n - 1, -1, -1):  # TypeProcessor.Remove: the last constraint takes the removed one's index
                index = first + i * stride
                lanes.append((refs[index], pre[index]))
                ops.append((bi, t, index, None))
                refs[index], pre[index] = refs[-1], pre[-1]
                refs.pop(); pre.pop()
                held[lanes[-1][0][lanes[-1][0] < (1 << 30)]] = False
            for i, (r, p) in enumerate(lanes):
                r = r.copy()
                if i % new_pairs == 0 and r[0] < (1 << 30) and r[1] < (1 << 30):
                    coming = {int(x) for q in range(i + 1, n) for x in lanes[q][0]}
                    for step in range(1, 17):
                        near = int(r[0]) + ((step + 1) // 2 if step & 1 else -(step // 2))
                        if 0 <= near < scene.body_count and near != r[1] and not held[near] and near not in coming and constrained[near]:
                            r[1] = near
                            break
                ops.append((bi, t, r, p))
                refs.append(r); pre.append(p)
                held[r[r < (1 << 30)]] = True
        return ops

    def apply(ops):  # ONE bepuhip_apply_structural_ops call (round 4); BATCHED=0: round 3's call per operation through ctypes
        if os.environ.get("BATCHED", "1") == "0":
            for bi, t, what, p in ops:
                if p is None:
                    solver.remove_constraint(bi, t, what)
                else:
                    solver.add_constraint(bi, t, what, p)
            return 0.0
        rows, words, at = [], [], 0
        for bi, t, what, p in ops:
            if p is None:
                rows.append((1, bi, t, what, 0, 0, 0, 0))
            else:
                rows.append((0, bi, t, -1, 0, 0, at, 0))
                words += [np.ascontiguousarray(what, dtype=np.int32).view(np.uint32), np.ascontiguousarray(p, dtype=np.float32).view(np.uint32)]
                at += what.size + p.size
        table, payload = np.asarray(rows, dtype=np.int32), np.concatenate(words)
        t0 = time.perf_counter()
        solver.apply_structural_op_table(table, payload)
        return 1e3 * (time.perf_counter() - t0)

    constrained = np.zeros(scene.body_count + 1, dtype=bool)
    for held in in_batch:
        constrained |= held
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    apply(churn()); solver.solve(1 / 60, sd, cb)
    frame_ms = calls_ms = call_ms = 0.0
    for _ in range(frames):
        ops = churn()
        t0 = time.perf_counter()
        call_ms += apply(ops) / frames
        t1 = time.perf_counter()
        solver.solve(1 / 60, sd, cb)  # flushes the updates, then solves
        calls_ms += 1e3 * (t1 - t0) / frames
        frame_ms += 1e3 * (time.perf_counter() - t1) / frames
    t0 = time.perf_counter()
    for _ in range(50):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    solve_ms = 1e3 * (time.perf_counter() - t0) / 50
    finite = bool(np.isfinite(solver.get_bodies(scene.body_count)).all())
    print(f"{name}, {label}: {calls} structural operations per frame {calls_ms:.2f} ms with the table built in Python, {call_ms:.2f} ms inside the one bepuhip_apply_structural_ops call; the flush + solve that follows {frame_ms:.3f} ms; the solve alone afterwards {solve_ms:.4f} ms; "
          f"clusters {solver.cluster_cycles().size}; finite {finite}", flush=True)
    if solver.schedule() == 0:  # what a re-plan costs and gives back: in the background with the churn going on (round 6), then bepuhip_replan itself
        for k in env:  # (the developer switch that made the updates leave the plan is off again: the new plan takes the operations of the frames in between)
            os.environ.pop(k, None)
        t0 = time.perf_counter()
        solver.replan_begin()
        begin_ms = 1e3 * (time.perf_counter() - t0)
        in_flight, slowest = 0, 0.0
        while True:
            t0 = time.perf_counter()
            committed = solver.replan_commit(wait=False)
            commit_ms = 1e3 * (time.perf_counter() - t0)
            if committed:
                break
            ops = churn()
            t0 = time.perf_counter()
            apply(ops)
            solver.solve(1 / 60, sd, cb)
            slowest = max(slowest, 1e3 * (time.perf_counter() - t0))
            in_flight += 1
        t0 = time.perf_counter()
        solver.solve(1 / 60, sd, cb)
        first_ms = 1e3 * (time.perf_counter() - t0)
        print(f"{name}: bepuhip_replan_begin {begin_ms:.2f} ms, {in_flight} frames of churn solved while the worker planned (slowest {slowest:.2f} ms), bepuhip_replan_commit {commit_ms:.2f} ms "
              f"({in_flight * calls} logged operations replayed) -> schedule {solver.schedule()}, first solve after it {first_ms:.3f} ms", flush=True)
        t0 = time.perf_counter()
        solver.replan()
        replan_ms = 1e3 * (time.perf_counter() - t0)
        for _ in range(20):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.sync()
        print(f"{name}: bepuhip_replan {replan_ms:.1f} ms -> schedule {solver.schedule()}, solve {1e3 * (time.perf_counter() - t0) / 50:.4f} ms", flush=True)
    solver.close()
    for k in env:
        os.environ.pop(k, None)
