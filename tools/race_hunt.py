"""Statistics on one nondeterministic device-vs-oracle mismatch (GPU box; a developer tool): the scene of a tools/fuzz_device.py ordinal is run `repeats` times under each of
a list of variations (environment switches of the library, solve parameters), and for every variation the number of runs that differ from the oracle is printed together
with the bodies that differ in the runs with the fewest differences (the closer to the origin of the fault, the fewer).
    python tools/race_hunt.py <seed> <ordinal> <repeats> [variation names ...]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import parity_util as pu
import small_scenes
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription
import replay_fuzz_device as rf  # parameters()

seed, ordinal, repeats = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
only = sys.argv[4:]
p0 = rf.parameters(seed, ordinal + 1)[ordinal]


def variant(**changes):
    p = dict(p0)
    cbk = {}
    for k, v in changes.items():
        if k in ("mode", "ivk", "asu"):
            cbk[k] = v
        else:
            p[k] = v
    if cbk:
        cb = p0["cb"]
        p["cb"] = PoseIntegratorCallbacks(gravity=cb.gravity, linear_damping=cb.linear_damping, angular_damping=cb.angular_damping,
                                          integrate_velocity_for_kinematics=cbk.get("ivk", cb.integrate_velocity_for_kinematics),
                                          allow_substeps_for_unconstrained_bodies=cbk.get("asu", cb.allow_substeps_for_unconstrained_bodies),
                                          angular_integration_mode=cbk.get("mode", cb.angular_integration_mode))
    return p


VARIATIONS = {
    "default": ({}, variant()),
    "threads64": ({"BEPUHIP_CLUSTER_THREADS": "64"}, variant()),
    "threads256": ({"BEPUHIP_CLUSTER_THREADS": "256"}, variant()),
    "threads512": ({"BEPUHIP_CLUSTER_THREADS": "512"}, variant()),
    "threads768": ({"BEPUHIP_CLUSTER_THREADS": "768"}, variant()),
    "mode0": ({}, variant(mode=0)),
    "mode1": ({}, variant(mode=1)),
    "frames1": ({}, variant(frames=1)),
    "sub1": ({}, variant(sub=1, its=[2])),
    "its1": ({}, variant(its=[1, 1])),
    "graph": ({}, variant(use_graph=True)),
    "ivk": ({}, variant(ivk=True)),
    "nokin": ({}, variant(kin=0.0)),
    "onecluster": ({"BEPUHIP_CLUSTER_BODIES": "4000"}, variant()),
    "batchpath": ({"BEPUHIP_NO_CLUSTERS": "1"}, variant()),
    "conserving_batchpath": ({"BEPUHIP_CONSERVING_CLUSTERS": "0"}, variant()),
    "policy0": ({"BEPUHIP_ROW_POLICY": "0"}, variant()),
    "jitter": ({"BEPUHIP_DEBUG_JITTER": "round"}, variant()),  # schedule fuzzing, a new seed every round
    "jitter_mode0": ({"BEPUHIP_DEBUG_JITTER": "round"}, variant(mode=0)),
    "jitter_mode1": ({"BEPUHIP_DEBUG_JITTER": "round"}, variant(mode=1)),
}
for drop in p0["types"]:
    VARIATIONS[f"without{drop}"] = ({}, variant(types=[t for t in p0["types"] if t != drop]))

print("scene:", rf.describe(p0), flush=True)
cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
cases = []
for name, (env, p) in VARIATIONS.items():
    if only and name not in only:
        continue
    scene = small_scenes.random_graph_scene(p["seed"], p["nb"], p["nc"], p["types"], kinematic_fraction=p["kin"])
    its = p["its"]
    sd = SolveDescription(1, p["sub"], velocity_iteration_scheduler=lambda s, its=its: its[s])
    refs = [scene.copy()]
    for _ in range(p["frames"]):
        refs.append(pu.run_oracle(refs[-1], 1 / 60, sd, p["cb"], frames=1, threads=4))
    cases.append({"name": name, "env": env, "p": p, "scene": scene, "sd": sd, "refs": refs, "bad": 0, "runs": 0, "first_frame": [], "detail": 0})
os.environ.pop("BEPUHIP_SPLIT_CLUSTERS", None)


def run_frames(case, round_index=0):
    """One upload, then frame by frame against the oracle's frames: returns the first frame that differs (or -1) and the downloaded state there."""
    env, p = {k: (str(round_index + 1) if v == "round" else v) for k, v in case["env"].items()}, case["p"]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        solver = HipSolver(use_clusters=p["use_clusters"], use_graph=p["use_graph"])
        s = case["scene"].copy()
        solver.upload(s, case["sd"].fallback_batch_threshold)
        out = (-1, None)
        for f in range(p["frames"]):
            solver.solve(1 / 60, case["sd"], p["cb"])
            solver.download(s)
            if not rf.exact(case["refs"][f + 1], s):
                out = (f, s.copy())
                break
        solver.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


for round_index in range(repeats):
    for case in cases:
        frame, got = run_frames(case, round_index)
        case["runs"] += 1
        if frame < 0:
            continue
        case["bad"] += 1
        case["first_frame"].append((round_index, frame))
        if case["detail"] < 4:
            case["detail"] += 1
            print(f"--- {case['name']} round {round_index}: first differing frame {frame}", flush=True)
            rf.report(case["refs"][frame + 1], got)
for case in cases:
    print(f"{case['name']:24s} {case['bad']:3d} of {case['runs']} runs differ; (round, first differing frame) {case['first_frame'][:12]} batches {len(case['scene'].batches)}", flush=True)
