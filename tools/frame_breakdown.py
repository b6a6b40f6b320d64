"""Where the resident frame's time goes: every stage of bench.py's `frame_through_abi_ms` frame on its own, each followed by a sync, then the
frame as the shim runs it. Host buffers registered. Developer tool, not part of the product."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks  # noqa: E402

scene, sd = bench.build_scene(int(os.environ.get("RAGDOLLS", "15000")), 5)
cb = PoseIntegratorCallbacks()
solver = HipSolver(device=0)
work = scene.copy()
solver.register_host_memory(work.bodies)
for b in work.batches:
    for tb in b:
        if tb.count:
            solver.register_host_memory(tb.prestep)
            solver.register_host_memory(tb.accumulated)
solver.upload(work)
contact = [(bi, tb) for bi, b in enumerate(work.batches) for tb in b if tb.count and TYPE_TABLE[tb.type_id][3].startswith("Contact")]
joints = [(bi, tb) for bi, b in enumerate(work.batches) for tb in b if tb.count and not TYPE_TABLE[tb.type_id][3].startswith("Contact")]
for _ in range(50):
    solver.solve(1 / 60, sd, cb, asynchronous=True)
solver.sync()
N = 20


def timed(f):
    f(); solver.sync()
    t0 = time.perf_counter()
    for _ in range(N):
        f()
        solver.sync()
    return 1e3 * (time.perf_counter() - t0) / N


def upd():
    for bi, tb in contact:
        solver.update_prestep(bi, tb.type_id, 0, tb.prestep, asynchronous=True)


def upd_joint_prestep():
    for bi, tb in joints:
        solver.update_prestep(bi, tb.type_id, 0, tb.prestep, asynchronous=True)


def frame():
    upd()
    solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.get_poses_and_velocities(work.bodies, asynchronous=True)


print(f"contact type batches {len(contact)} ({sum(tb.prestep.nbytes for _, tb in contact) / 1e6:.1f} MB prestep), joint type batches {len(joints)} "
      f"({sum(tb.prestep.nbytes for _, tb in joints) / 1e6:.1f} MB prestep, {sum(tb.accumulated.nbytes for _, tb in joints) / 1e6:.1f} MB impulses)")
print(f"sync alone                         {timed(lambda: None):8.3f} ms")
print(f"update_prestep_async x contacts    {timed(upd):8.3f} ms")
print(f"update_prestep_async x joints      {timed(upd_joint_prestep):8.3f} ms")
print(f"solve_async                        {timed(lambda: solver.solve(1 / 60, sd, cb, asynchronous=True)):8.3f} ms")
print(f"get_poses_and_velocities_async     {timed(lambda: solver.get_poses_and_velocities(work.bodies, asynchronous=True)):8.3f} ms")
print(f"set_bodies                         {timed(lambda: solver.set_bodies(work.bodies)):8.3f} ms")
print(f"whole frame                        {timed(frame):8.3f} ms")
t0 = time.perf_counter()
for _ in range(N):
    for bi, tb in joints:
        solver.lib.bepuhip_get_accumulated_impulses(solver.ctx, bi, tb.type_id, tb.accumulated.ctypes.data_as(__import__('ctypes').POINTER(__import__('ctypes').c_float)))
print(f"get_accumulated_impulses x joints  {1e3 * (time.perf_counter() - t0) / N:8.3f} ms (synchronous calls)")
solver.close()
