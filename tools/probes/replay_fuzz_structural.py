"""Replays tools/fuzz_structural.py's scene stream (GPU box; developer tool). The generator is one random stream, so the scenes before the wanted one have to run too — once:
    python tools/probes/replay_fuzz_structural.py <seed> <scenes>                      runs scenes [0, scenes) and prints every mismatch report
    python tools/probes/replay_fuzz_structural.py <seed> <ordinal> --save <file>       runs scenes [0, ordinal) and saves the generator's state in front of scene <ordinal>
    python tools/probes/replay_fuzz_structural.py <seed> <ordinal> --load <file>       runs scene <ordinal> alone from a saved state (a second; environment switches apply)"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import fuzz_util as fu

seed, count = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else None
rng = np.random.default_rng(seed)


def jitter_of(scene):
    return ((seed * 7919 + scene) | 1) if scene % 2 else 0


if mode == "--load":
    state = json.load(open(sys.argv[4]))
    rng.bit_generator.state = state
    stats = fu.run_structural_scene(rng, jitter=jitter_of(count))
    print(f"scene {count} alone:", "ok" if stats["ok"] else stats["report"], {k: v for k, v in stats.items() if k != "report"}, flush=True)
    sys.exit(0)
bad = 0
for scene in range(count):
    stats = fu.run_structural_scene(rng, jitter=jitter_of(scene))
    if not stats["ok"]:
        bad += 1
        print(f"scene {scene}:", stats["report"], flush=True)
if mode == "--save":
    state = rng.bit_generator.state
    json.dump(state, open(sys.argv[4], "w"), default=int)
    print(f"state in front of scene {count} saved to {sys.argv[4]}")
print(f"library {os.environ.get('BEPUHIP_LIB', 'product')}: {count} scenes of seed {seed}, mismatching {bad}")
