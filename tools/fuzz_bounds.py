"""Fuzzer of the two CPU restatements of PredictBoundingBoxes against each other (CPU only; a developer tool): random tables of hulls, compounds and meshes, bodies of all
nine shape types with random kinematic fractions, velocity scales from rest to fast spin, callbacks and time steps, oracle/bepu_bounds.h against oracle/wide/wide_bounds.h
bit for bit.   python tools/fuzz_bounds.py [seconds]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import test_bounds as tb, oracle_ffi, wide_ffi, small_scenes
from bepuphysics2_amd.scene import PoseIntegratorCallbacks
rng=np.random.default_rng(5)
t_end=time.time()+(float(sys.argv[1]) if len(sys.argv)>1 else 60); n=bad=0
while time.time()<t_end:
    hulls=tb._random_hulls(rng,int(rng.integers(1,12))); meshes=tb._random_meshes(rng,int(rng.integers(1,8)))
    compounds=tb._random_compounds(rng,int(rng.integers(1,20)),len(hulls))
    nb=int(rng.integers(1,700))
    bodies=tb._spinning_bodies(rng,nb)
    for i in range(nb):
        if rng.random()<0.2: bodies[i]=small_scenes.kinematic_body(rng, rng.uniform(-5,5,3), angular=tuple(rng.uniform(-2,2,3)))
    if rng.random()<0.3: bodies[:,8:15]*=float(rng.choice([0,1e-3,30]))
    coll=tb._every_shape_collidables(rng,nb,len(hulls),len(compounds),len(meshes))
    cb=PoseIntegratorCallbacks(gravity=tuple(rng.uniform(-10,10,3)), linear_damping=float(rng.uniform(0,0.9)), angular_damping=float(rng.uniform(0,0.9)), integrate_velocity_for_kinematics=bool(rng.integers(2)))
    dt=float(rng.choice([1/60,1/240,0.1]))
    a=oracle_ffi.predict_bounding_boxes(bodies,dt,cb,coll,hulls,compounds,meshes)
    b=wide_ffi.predict_bounding_boxes(bodies,dt,cb,coll,hulls,compounds,meshes)
    n+=1
    if not np.array_equal(a.view(np.int32),b.view(np.int32)):
        bad+=1; d=np.flatnonzero((a.view(np.int32).reshape(nb,8)!=b.view(np.int32).reshape(nb,8)).any(axis=1)); print("MISMATCH",nb,dt,d[:5],coll["shape_type"][d[:5]],a[d[:2]],b[d[:2]],flush=True)
print("bounds scenes",n,"mismatches",bad)
