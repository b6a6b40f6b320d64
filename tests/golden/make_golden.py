"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference (C#) cannot run here and its tests hold no golden vectors for this path (parity unpinned), so these
fixtures are regression pins of OUR restatement: (a) the reference's own microbenchmark inputs
(DemoBenchmarks/TwoBodyConstraintBenchmarks.cs:42-117 — identity orientations/inertias, B at (2,0,0), dt = 1/60, 1000 x WarmStart+Solve),
for which the answer is analytically zero, plus perturbed variants; (b) small seeded scenes after a few frames."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import oracle_ffi  # noqa: E402
import small_scenes  # noqa: E402
from bepuphysics2_amd.scene import HOT_PATH_TYPES, TYPE_TABLE, WIDENED_TYPES, PoseIntegratorCallbacks, SolveDescription, make_body  # noqa: E402

SPRING = [np.float32(20 * np.float32(np.pi)), np.float32(2)]  # AngularFrequency = 20*pi, TwiceDampingRatio = 2


def microbench_inputs():
    """(name, type_id, prestep lane) exactly as TwoBodyConstraintBenchmarks.cs:44-62,92-116."""
    contact4 = []
    for _ in range(4):
        contact4 += [1, 0, 0, 0]  # OffsetA (1,0,0), Depth 0
    contact4 += [2, 0, 0] + [0, 1, 0] + [1.0] + SPRING + [2.0]
    ball = [1, 0, 0, -1, 0, 0] + SPRING
    hinge = [0, 1, 0, 0, 1, 0] + SPRING
    return [("Contact4", 7, contact4), ("BallSocket", 22, ball), ("AngularHinge", 23, hinge)]


def run_micro(type_id, lane, vel_a=None, vel_b=None, iterations=1000):
    a = make_body(position=(0, 0, 0))
    b = make_body(position=(2, 0, 0))
    for body in (a, b):
        body[24:30] = body[16:22]  # world inverse inertia = identity too
        body[30] = body[22]
    if vel_a is not None:
        a[8:11], a[12:15] = vel_a[:3], vel_a[3:]
    if vel_b is not None:
        b[8:11], b[12:15] = vel_b[:3], vel_b[3:]
    p = np.asarray(lane, dtype=np.float32).copy()
    acc = np.zeros(TYPE_TABLE[type_id][2], dtype=np.float32)
    oracle_ffi.constraint_iterate(type_id, a, b, p, acc, 1.0 / 60.0, iterations)
    return a, b, acc


def main():
    out = {}
    for name, type_id, lane in microbench_inputs():
        a, b, acc = run_micro(type_id, lane)
        out[f"micro_{name}_a"], out[f"micro_{name}_b"], out[f"micro_{name}_acc"] = a, b, acc
        va = np.asarray([0.3, -0.2, 0.1, 0.05, 0.4, -0.3], np.float32)
        vb = np.asarray([-0.1, 0.25, 0.0, -0.2, 0.1, 0.15], np.float32)
        a, b, acc = run_micro(type_id, lane, va, vb, iterations=8)
        out[f"microv_{name}_a"], out[f"microv_{name}_b"], out[f"microv_{name}_acc"] = a, b, acc
    np.savez_compressed(os.path.join(HERE, "microbench.npz"), **out)

    scenes = {}
    sd, cb = SolveDescription(2, 8), PoseIntegratorCallbacks()
    for seed, types in ((1, HOT_PATH_TYPES), (2, [0, 1, 2, 3, 4, 5, 6, 7]), (3, [22, 23, 25, 26, 27, 30, 46, 47])):
        sc = small_scenes.random_graph_scene(seed, 120, 300, types)
        for _ in range(2):
            oracle_ffi.solve(sc, 1 / 60, sd, cb)
        scenes[f"graph{seed}_bodies"] = sc.bodies
        scenes[f"graph{seed}_impulses"] = np.concatenate([tb.accumulated_lanes().reshape(-1) for b in sc.batches for tb in b])
    st = small_scenes.box_stack_scene()
    for _ in range(4):
        oracle_ffi.solve(st, 1 / 60, SolveDescription(4, 1), cb)
    scenes["stack_bodies"] = st.bodies
    np.savez_compressed(os.path.join(HERE, "small_scenes.npz"), **scenes)
    widened = {}
    for type_id in WIDENED_TYPES:  # one fixture per added type
        sc = small_scenes.random_graph_scene(400 + type_id, 120, 300, [type_id])
        for _ in range(2):
            oracle_ffi.solve(sc, 1 / 60, sd, cb)
        widened[f"type{type_id}_bodies"] = sc.bodies
    np.savez_compressed(os.path.join(HERE, "widened_types.npz"), **widened)
    bounds()
    print("wrote", os.listdir(HERE))


def bounds_inputs():
    """The seeded body set of the PredictBoundingBoxes fixture: all nine shape types, fast spinners, runs of kinematic bodies, a ragged last bundle."""
    import test_bounds as tb
    rng = np.random.default_rng(101)
    hulls, meshes = tb._random_hulls(rng, 10), tb._random_meshes(rng, 6)
    compounds = tb._random_compounds(rng, 14, len(hulls))
    n = 603
    bodies = tb._spinning_bodies(rng, n)
    for start in (40, 200, 413):
        for i in range(start, start + 19):
            bodies[i] = small_scenes.kinematic_body(rng, rng.uniform(-5, 5, 3), angular=(0.3, -1.2, 0.4))
    coll = tb._every_shape_collidables(rng, n, len(hulls), len(compounds), len(meshes))
    return bodies, coll, hulls, compounds, meshes


def bounds():
    bodies, coll, hulls, compounds, meshes = bounds_inputs()
    out = {}
    for name, cb in (("default", PoseIntegratorCallbacks()),
                     ("kinematics_integrated", PoseIntegratorCallbacks(gravity=(1, -9, 0.5), linear_damping=0.1, angular_damping=0.2, integrate_velocity_for_kinematics=True))):
        r = oracle_ffi.predict_bounding_boxes(bodies, 1 / 60, cb, coll, hulls, compounds, meshes)
        out[name] = r.view(np.int32).reshape(-1, 8)
    np.savez_compressed(os.path.join(HERE, "bounds.npz"), **out)


if __name__ == "__main__":
    if "--bounds-only" in sys.argv:
        bounds()
    else:
        main()
