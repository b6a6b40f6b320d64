"""How long the first solves of a process take (round 5: bench.py --lattice measured 0.78 ms/step over its steps 10-60 and 0.26 after 300 pre-warm steps). Every solve of
the first 150 timed on its own (launch + sync), with the launch policy measured as usual and pinned to plain rows, on the headline scene and on the connected lattice.
Developer probe, not part of the product."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import PoseIntegratorCallbacks  # noqa: E402

for name, args in (("ragdoll tube", ("ragdoll_tube", 15000, 1, 0, 5)), ("lattice", ("ragdoll_tube", 15000, 1, 1, 5))):
    sim = HostSimulation.scene(*args)
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    for policy in (None, "0"):
        if policy is None:
            os.environ.pop("BEPUHIP_ROW_POLICY", None)
        else:
            os.environ["BEPUHIP_ROW_POLICY"] = policy
        os.environ["BEPUHIP_POLICY_CACHE"] = "0"
        solver = HipSolver(exclusive_device=True)
        solver.upload(scene)
        cb = PoseIntegratorCallbacks()
        times = []
        for _ in range(150):
            t0 = time.perf_counter()
            solver.solve(1 / 60, sd, cb)
            times.append(1e3 * (time.perf_counter() - t0))
        solver.close()
        groups = [sum(times[i:i + 10]) / 10 for i in range(0, 150, 10)]
        print(f"{name:13s} policy {'measured' if policy is None else 'plain rows'}: first solve {times[0]:.3f} ms, then means of ten: " + " ".join(f"{g:.3f}" for g in groups), flush=True)
