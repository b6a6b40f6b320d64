"""Fuzzer of the two CPU restatements against each other (CPU only; a developer tool, not part of the product or of the test suite).

Random constraint graphs over random subsets of the 44 type ids, random body / constraint counts (small body counts with many constraints reach the sequential fallback
batch), kinematic fractions, substep counts with uneven iteration schedules, integrator options, angular modes and worker counts: two frames through oracle/ and
through oracle/wide, compared bit for bit (bodies, accumulated impulses, prestep data of the occupied lanes).   python tools/fuzz_oracles.py <seed> <seconds>"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import oracle_ffi, wide_ffi, small_scenes
from small_scenes import TYPE_TABLE
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription
ALL=sorted(TYPE_TABLE.keys())
def bits(a): return np.ascontiguousarray(a).view(np.int32)
COLS=[0,1,2,3,4,5,6,8,9,10,12,13,14,16,17,18,19,20,21,22,24,25,26,27,28,29,30]
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
t_end=time.time()+float(sys.argv[2]) if len(sys.argv)>2 else time.time()+300
n=0; bad=0
while time.time()<t_end:
    seed=int(rng.integers(1<<30))
    k=int(rng.integers(1,12)); types=list(rng.choice(ALL,size=k,replace=False))
    nb=int(rng.integers(20,400)); nc=int(rng.integers(10,1200))
    kin=float(rng.choice([0,0.05,0.3]))
    try:
        scene=small_scenes.random_graph_scene(seed, nb, nc, [int(t) for t in types], kinematic_fraction=kin)
    except Exception as e:
        continue
    sub=int(rng.integers(1,6)); its=[int(x) for x in rng.integers(1,4,size=sub)]
    sd=SolveDescription(1, sub, velocity_iteration_scheduler=lambda s: its[s])
    cb=PoseIntegratorCallbacks(gravity=tuple(rng.uniform(-10,10,3)), linear_damping=float(rng.uniform(0,0.5)), angular_damping=float(rng.uniform(0,0.5)),
        integrate_velocity_for_kinematics=bool(rng.integers(2)), allow_substeps_for_unconstrained_bodies=bool(rng.integers(2)), angular_integration_mode=int(rng.integers(3)))
    a,b=scene.copy(),scene.copy()
    th=int(rng.choice([1,1,3,4]))
    for f in range(2):
        oracle_ffi.solve(a,1/60,sd,cb); wide_ffi.solve(b,1/60,sd,cb,threads=th)
    ok=np.array_equal(bits(a.bodies[:,COLS]),bits(b.bodies[:,COLS]))
    for ba,bb in zip(a.batches,b.batches):
        for ta,tb in zip(ba,bb):
            occ=ta.occupied()
            ok&=np.array_equal(bits(ta.accumulated_lanes()[occ]),bits(tb.accumulated_lanes()[occ])) and np.array_equal(bits(ta.prestep_lanes()[occ]),bits(tb.prestep_lanes()[occ]))
    n+=1
    if not ok:
        bad+=1; print("MISMATCH seed",seed,types,nb,nc,kin,sub,its,th,cb, flush=True)
print("scenes",n,"mismatches",bad)
