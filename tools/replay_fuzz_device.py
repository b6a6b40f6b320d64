"""Replays scenes of a tools/fuzz_device.py run by their ordinal (the fuzzer's parameters come from one seeded generator, independent of the device), device against oracle,
and says WHAT differs when something does (which field, how many words, whether NaN is involved). A developer tool (GPU box).
    python tools/replay_fuzz_device.py <seed> <ordinal,ordinal,...> [seconds for the remaining ordinals] [log file]
(ordinals of tools/fuzz_device.py since round 4 run odd ordinals under schedule fuzzing with jitter seed (seed * 7919 + ordinal) | 1: replayed the same way when JITTER=1)
Every scene's parameters are appended to the log file BEFORE its device run (a hang or a cut output still leaves them). A mismatching scene is triaged on the spot:
replayed several times as it is (does it reproduce? are the device's runs equal to each other?) and once under every developer switch that takes one of the
library's optional paths out."""
import json
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

import fuzz_util as fu
from fuzz_util import describe, exact  # noqa: F401  (tools/race_hunt.py uses them under these names)

LOG = None
parameters = fu.device_scene_parameters

SWITCHES = [{}, {}, {}, {"BEPUHIP_ROW_POLICY": "0"}, {"BEPUHIP_ROW_POLICY": "1"}, {"BEPUHIP_ROW_POLICY": "2"}, {"BEPUHIP_CONSERVING_CLUSTERS": "0"},
            {"BEPUHIP_SPLIT_MANY_BODY": "0"}, {"BEPUHIP_SPLIT_LOCAL_HANDOFF": "0"}, {"BEPUHIP_NO_SPLIT": "1"}, {"BEPUHIP_NO_CLUSTERS": "1"}, {"BEPUHIP_COOPERATIVE": "0"},
            {"BEPUHIP_CLUSTER_THREADS": "512", "BEPUHIP_SPLIT_THREADS": "1024"}]


def log(text):
    print(text, flush=True)
    if LOG:
        LOG.write(text + "\n"); LOG.flush(); os.fsync(LOG.fileno())


def device(p, scene, sd, env, jitter=0):
    return fu.run_device(p, scene, sd, env=env, jitter=jitter)


def report(ref, got):
    cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
    rb, gb = ref.bodies[:, cols], got.bodies[:, cols]
    diff = rb.view(np.int32) != gb.view(np.int32)
    rows = np.nonzero(diff.any(axis=1))[0]
    log(f"  bodies: {int(diff.sum())} words differ in {rows.size} bodies {rows[:12].tolist()}; NaN in the oracle's {int(np.isnan(rb).sum())}, in the device's {int(np.isnan(gb).sum())}; "
        f"differing words where both are NaN {int((diff & np.isnan(rb) & np.isnan(gb)).sum())}; max |oracle| {float(np.nanmax(np.abs(rb))):.4g}")
    for r in rows[:3]:
        log(f"    body {int(r)}: oracle {rb[r].tolist()} device {gb[r].tolist()}")
    w = ref.bundle_width
    for bi, (br, bg) in enumerate(zip(ref.batches, got.batches)):
        for tr, tg in zip(br, bg):
            occupied = tr.occupied(w)
            for name, ar, ag in (("impulses", tr.accumulated_lanes(w)[occupied], tg.accumulated_lanes(w)[occupied]), ("prestep", tr.prestep_lanes(w)[occupied], tg.prestep_lanes(w)[occupied])):
                d = ar.view(np.int32) != ag.view(np.int32)
                if d.any():
                    both_nan = d & np.isnan(ar) & np.isnan(ag)
                    log(f"  batch {bi} type {tr.type_id} {name}: {int(d.sum())} words differ in {int(d.any(axis=1).sum())} of {tr.count} constraints {np.nonzero(d.any(axis=1))[0][:8].tolist()}; both NaN in {int(both_nan.sum())} of them; "
                        f"first: oracle {ar[d][:3]} device {ag[d][:3]} bits {[hex(int(x)) for x in ar.view(np.uint32)[d][:3]]} / {[hex(int(x)) for x in ag.view(np.uint32)[d][:3]]}")


def run(index, p, jitter=0):
    scene, sd = fu.build_device_scene(p)
    if LOG:
        LOG.write(f"ordinal {index}: {describe(p)} batches {len(scene.batches)}\n"); LOG.flush(); os.fsync(LOG.fileno())
    ref = pu.run_oracle(scene, 1 / 60, sd, p["cb"], frames=p["frames"], threads=4)
    got, info = device(p, scene, sd, {}, jitter)
    if exact(ref, got):
        return True
    log(f"ordinal {index}: MISMATCH {describe(p)} batches {len(scene.batches)} schedule/policy/clusters {info}")
    finite = bool(np.isfinite(ref.bodies[:, :15]).all())
    log(f"  oracle bodies finite: {finite}")
    report(ref, got)
    previous = got
    for env in SWITCHES:
        again, info2 = device(p, scene, sd, env, jitter)
        same_as_first = pu.compare_scenes(got, again)
        log(f"  again with {env or 'nothing changed'}: equal to the oracle {exact(ref, again)}, equal to the first device run {same_as_first['bodies_bit_exact'] and same_as_first['impulses_bit_exact']}, schedule/policy/clusters {info2}")
        previous = again
    return False


if __name__ == "__main__":
    first = [int(x) for x in sys.argv[2].split(",") if x] if len(sys.argv) > 2 else []
    t_end = time.time() + (float(sys.argv[3]) if len(sys.argv) > 3 else 0)
    LOG = open(sys.argv[4], "a") if len(sys.argv) > 4 else None
    params = parameters(int(sys.argv[1]), 600)
    done = set()
    bad = 0
    fuzz_seed = int(sys.argv[1])
    jitter_of = (lambda ordinal: ((fuzz_seed * 7919 + ordinal) | 1) if ordinal % 2 else 0) if os.environ.get("JITTER") else (lambda ordinal: 0)
    for index in first:
        bad += not run(index, params[index], jitter_of(index)); done.add(index)
    index = 0
    while time.time() < t_end and index < len(params):
        if index not in done:
            bad += not run(index, params[index], jitter_of(index)); done.add(index)
        index += 1
    log(f"replayed {len(done)} scenes, mismatches {bad}")
