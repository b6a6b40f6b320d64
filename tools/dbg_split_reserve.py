import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import parity_util as pu
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
for args in (("pile", 8000, 0, 0, 5), ("ragdoll_tube", 1200, 1, 2, 11)):
    sim = HostSimulation.scene(*args); scene, sd = sim.export(), sim.solve_description(); sim.close()
    cb = PoseIntegratorCallbacks()
    ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=3, threads=4)
    for reserve in (False, True):
        s = HipSolver(reserve_update_slots=reserve); got = pu.run_hip(s, scene, 1 / 60, sd, cb, frames=3); n = s.cluster_cycles().size; s.close()
        m = pu.compare_scenes(ref, got)
        print(args[0], "reserve", reserve, "clusters", n, "exact", m["bodies_bit_exact"], m["impulses_bit_exact"], m["prestep_bit_exact"])
