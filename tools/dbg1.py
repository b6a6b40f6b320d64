import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.hostlib import HostSimulation
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sim = HostSimulation.scene("ragdoll_tube", n, 1, 0, 5); scene, sd = sim.export(), sim.solve_description(); sim.close()
print("scene", scene.body_count, flush=True)
s = HipSolver()
print("created", flush=True)
s.set_bodies(scene.bodies)
print("set_bodies ok", flush=True)
s.upload(scene)
print("upload ok", flush=True)
s.solve(1/60, sd, PoseIntegratorCallbacks())
print("solve ok", flush=True)
s.close()
