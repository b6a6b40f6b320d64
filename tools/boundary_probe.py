"""Timing of the upload path (begin / set_type_batch / end) and of the resident frame on the bench scene. Not part of the product."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
scene, sd = bench.build_scene(int(os.environ.get("RAGDOLLS", "15000")), 5)
out = bench.boundary_leg(scene, sd, PoseIntegratorCallbacks(), 0)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if k.endswith("_ms")})
