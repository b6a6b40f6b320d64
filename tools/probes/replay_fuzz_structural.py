"""Replays tools/fuzz_structural.py's scene stream of a seed up to a scene ordinal and prints every mismatch report (GPU box; developer tool): the generator is one random
stream, so the scenes before the wanted one have to run too.   python tools/probes/replay_fuzz_structural.py <seed> <scenes>"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import fuzz_util as fu

seed, count = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for scene in range(count):
    stats = fu.run_structural_scene(rng, jitter=((seed * 7919 + scene) | 1) if scene % 2 else 0)
    if not stats["ok"]:
        bad += 1
        print(f"scene {scene}:", stats["report"], flush=True)
print(f"library {os.environ.get('BEPUHIP_LIB', 'product')}: {count} scenes of seed {seed}, mismatching {bad}")
