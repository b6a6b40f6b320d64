"""Same-box A/B of libraries on the 'rigs' scene (the bench scene's graph with seven joint types replaced by widened ones — tools/perf_widened.py), one child process per
library (BEPUHIP_LIB is read when the bindings load). Not part of the product.
    python tools/ab_rigs.py <label>=<path to libbepuhip.so or ''> ..."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription
from bepuphysics2_amd.synthetic import rig_scene
scene = rig_scene(15000)[0]
sd, cb = SolveDescription(1, 4), PoseIntegratorCallbacks()
s = HipSolver(exclusive_device=True)
s.upload(scene)
for _ in range(100):
    s.solve(1 / 60, sd, cb, asynchronous=True)
s.reset_state(); s.sync()
times = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        s.solve(1 / 60, sd, cb, asynchronous=True)
    s.sync()
    times.append(1e3 * (time.perf_counter() - t0) / 200)
b = s.get_bodies(scene.body_count)
import zlib
print(f"{min(times):.4f} ms/step (runs {', '.join(f'{t:.4f}' for t in times)}) kcycles max {int(s.cluster_cycles().max()) // 1000} family {s.kernel_family()} crc {zlib.crc32(b.tobytes()):08x}")
'''
for round_ in range(2):
    for spec in sys.argv[1:]:
        label, _, path = spec.partition("=")
        env = dict(os.environ)
        if path:
            env["BEPUHIP_LIB"] = path
        else:
            env.pop("BEPUHIP_LIB", None)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode().strip().splitlines()
        print(f"  {label:<40s} {out[-1] if out else '(no output)'}", flush=True)
