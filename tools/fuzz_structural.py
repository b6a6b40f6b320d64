"""Fuzzer of structural updates on the device against the oracle solving the host mirror (GPU box; a developer tool — tests/test_gpu_schedule_fuzz.py runs a fixed slice of
the same generator, tests/fuzz_util.py: run_structural_scene, in the GPU suite). Every second scene runs under schedule fuzzing (BEPUHIP_DEBUG_JITTER).
    python tools/fuzz_structural.py <seed> <seconds>"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import fuzz_util as fu

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
scenes = frames_total = bad = stayed = refused = split_scenes = split_stayed = replans = background = body_removals = fallback_scenes = 0
while time.time() < t_end:
    stats = fu.run_structural_scene(rng, jitter=((seed * 7919 + scenes) | 1) if scenes % 2 else 0)
    if not stats["ok"]:
        print(f"scene {scenes}:", stats["report"], flush=True)
    scenes += 1
    bad += not stats["ok"]
    frames_total += stats["frames"]; refused += stats["refused"]; replans += stats["replans"]; background += stats.get("background_replans", 0); body_removals += stats["body_removals"]
    stayed += stats["on_island_schedule"]; split_scenes += stats["big"]; split_stayed += stats["on_split_plan"]; fallback_scenes += stats.get("fallback_threshold", 64) < 64
print(f"re-plans {replans} + {background} in the background (bepuhip_replan_begin / _commit, with the frames' operations replayed), body removals {body_removals}; scenes with a low FallbackBatchThreshold (additions to / removals from the sequential fallback batch) {fallback_scenes}; scenes {scenes} ({split_scenes} big enough for a split-island plan, {split_stayed} of them still on it at the end), frames {frames_total}, still on an island schedule at the end {stayed}, "
      f"ended by a refused fallback-batch addition {refused}, mismatching scenes {bad}")
