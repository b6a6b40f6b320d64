import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import parity_util as pu
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
sim = HostSimulation.scene("ragdoll_tube", 1200, 1, 2, 11); scene, sd = sim.export(), sim.solve_description(); sim.close()
cb = PoseIntegratorCallbacks()
FRAMES = int(os.environ.get("FRAMES", "3"))
ref = pu.run_oracle(scene, 1 / 60, sd, cb, frames=FRAMES, threads=4)
for mode in ("0", "1", "2", "3", "4"):
    os.environ["BEPUHIP_SPLIT_LOCAL_HANDOFF"] = mode
    s = HipSolver(); got = pu.run_hip(s, scene, 1 / 60, sd, cb, frames=FRAMES); n = s.cluster_cycles().size; s.close()
    m = pu.compare_scenes(ref, got)
    bad = np.nonzero(np.any(ref.bodies[:, [8, 9, 10, 12, 13, 14]].view(np.int32) != got.bodies[:, [8, 9, 10, 12, 13, 14]].view(np.int32), axis=1))[0]
    lin = np.nonzero(np.any(ref.bodies[:, [8, 9, 10]].view(np.int32) != got.bodies[:, [8, 9, 10]].view(np.int32), axis=1))[0]
    print("mode", mode, "clusters", n, "exact", m["bodies_bit_exact"], "bad bodies", bad.size, "with wrong linear velocity", lin.size, "first", bad[:8], "vel err", m["velocity_rel_err"])
