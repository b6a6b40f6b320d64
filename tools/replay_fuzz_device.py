"""Replays scenes of a tools/fuzz_device.py run by their ordinal (the fuzzer's parameters come from one seeded generator, independent of the device), device against oracle,
and says WHAT differs when something does (which field, how many words, whether NaN is involved). A developer tool (GPU box).
    python tools/replay_fuzz_device.py <seed> <ordinal,ordinal,...> [seconds for the remaining ordinals]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import parity_util as pu
import small_scenes
from small_scenes import TYPE_TABLE
from bepuphysics2_amd.native import HipSolver
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

ALL = sorted(TYPE_TABLE.keys())
first = [int(x) for x in sys.argv[2].split(",") if x] if len(sys.argv) > 2 else []
t_end = time.time() + (float(sys.argv[3]) if len(sys.argv) > 3 else 0)


def parameters(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        p = {"seed": int(rng.integers(1 << 30)), "big": bool(rng.random() < 0.15)}
        p["types"] = [int(t) for t in rng.choice(ALL, size=int(rng.integers(1, 10)), replace=False)]
        p["nb"], p["nc"] = (int(rng.integers(3000, 7000)), int(rng.integers(6000, 16000))) if p["big"] else (int(rng.integers(20, 600)), int(rng.integers(10, 2500)))
        p["kin"] = float(rng.choice([0, 0.05, 0.3]))
        p["sub"] = int(rng.integers(1, 6))
        p["its"] = [int(x) for x in rng.integers(1, 4, size=p["sub"])]
        p["cb"] = PoseIntegratorCallbacks(gravity=tuple(rng.uniform(-10, 10, 3)), linear_damping=float(rng.uniform(0, 0.5)), angular_damping=float(rng.uniform(0, 0.5)),
                                          integrate_velocity_for_kinematics=bool(rng.integers(2)), allow_substeps_for_unconstrained_bodies=bool(rng.integers(2)),
                                          angular_integration_mode=int(rng.integers(3)))
        p["use_clusters"] = bool(rng.random() < 0.8)
        p["split_clusters"] = int(rng.integers(16, 40)) if p["big"] else None
        p["frames"] = int(rng.integers(1, 4))
        p["use_graph"] = bool(rng.integers(2))
        out.append(p)
    return out


def run(index, p):
    scene = small_scenes.random_graph_scene(p["seed"], p["nb"], p["nc"], p["types"], kinematic_fraction=p["kin"])
    its = p["its"]
    sd = SolveDescription(1, p["sub"], velocity_iteration_scheduler=lambda s: its[s])
    if p["split_clusters"]:
        os.environ["BEPUHIP_SPLIT_CLUSTERS"] = str(p["split_clusters"])
    else:
        os.environ.pop("BEPUHIP_SPLIT_CLUSTERS", None)
    ref = pu.run_oracle(scene, 1 / 60, sd, p["cb"], frames=p["frames"], threads=4)
    solver = HipSolver(use_clusters=p["use_clusters"], use_graph=p["use_graph"])
    got = pu.run_hip(solver, scene, 1 / 60, sd, p["cb"], frames=p["frames"])
    schedule = solver.schedule()
    solver.close()
    m = pu.compare_scenes(ref, got)
    if m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"]:
        return True
    print(f"ordinal {index}: MISMATCH seed {p['seed']} types {p['types']} bodies {p['nb']} constraints {p['nc']} substeps {p['sub']} x {its} mode {p['cb'].angular_integration_mode} "
          f"schedule {schedule} graph {p['use_graph']} frames {p['frames']} batches {len(scene.batches)}", flush=True)
    cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
    rb, gb = ref.bodies[:, cols], got.bodies[:, cols]
    diff = rb.view(np.int32) != gb.view(np.int32)
    print(f"  bodies: {int(diff.sum())} words differ in {int(diff.any(axis=1).sum())} bodies; NaN in the oracle's {int(np.isnan(rb).sum())}, in the device's {int(np.isnan(gb).sum())}; "
          f"differing words where both are NaN {int((diff & np.isnan(rb) & np.isnan(gb)).sum())}", flush=True)
    w = ref.bundle_width
    for bi, (br, bg) in enumerate(zip(ref.batches, got.batches)):
        for tr, tg in zip(br, bg):
            occupied = tr.occupied(w)
            for name, ar, ag in (("impulses", tr.accumulated_lanes(w)[occupied], tg.accumulated_lanes(w)[occupied]), ("prestep", tr.prestep_lanes(w)[occupied], tg.prestep_lanes(w)[occupied])):
                d = ar.view(np.int32) != ag.view(np.int32)
                if d.any():
                    both_nan = d & np.isnan(ar) & np.isnan(ag)
                    print(f"  batch {bi} type {tr.type_id} {name}: {int(d.sum())} words differ in {int(d.any(axis=1).sum())} of {tr.count} constraints; both NaN in {int(both_nan.sum())} of them; "
                          f"first: oracle {ar[d][:3]} device {ag[d][:3]} bits {[hex(int(x)) for x in ar.view(np.uint32)[d][:3]]} / {[hex(int(x)) for x in ag.view(np.uint32)[d][:3]]}", flush=True)
    return False


params = parameters(int(sys.argv[1]), 600)
done = set()
bad = 0
for index in first:
    bad += not run(index, params[index]); done.add(index)
index = 0
while time.time() < t_end and index < len(params):
    if index not in done:
        bad += not run(index, params[index]); done.add(index)
    index += 1
print(f"replayed {len(done)} scenes, mismatches {bad}")
