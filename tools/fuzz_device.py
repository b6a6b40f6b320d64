"""Fuzzer of the HIP path against the scalar oracle (GPU box; a developer tool — tests/test_gpu_schedule_fuzz.py runs a fixed slice of the same generator in the GPU suite).

Random constraint graphs over random subsets of the 44 type ids (sizes from a handful of constraints to split-island plans), kinematic fractions, substep counts with uneven
iteration schedules, integrator options, angular modes, both schedules, forced split plans: `frames` frames through oracle/ and through libbepuhip.so, compared bit for bit.
Every second scene runs under schedule fuzzing (BEPUHIP_DEBUG_JITTER, bepu_cluster_kernel.h: jitter_nap). The parameters of a scene are printed BEFORE its device run when a
log file is given, and a mismatch prints its ordinal: `tools/replay_fuzz_device.py <seed> <ordinal>` replays exactly that scene.
    python tools/fuzz_device.py <seed> <seconds> [log file]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import fuzz_util as fu
from bepuphysics2_amd.native import UnsupportedError

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60)
log = open(sys.argv[3], "a") if len(sys.argv) > 3 else None
params = fu.device_scene_parameters(seed, 20000)
n = bad = split = batch_path = refused = diverged = jittered = special = 0
for ordinal, p in enumerate(params):
    if time.time() >= t_end:
        break
    jitter = (seed * 7919 + ordinal) | 1 if ordinal % 2 else 0
    if log:
        log.write(f"ordinal {ordinal} jitter {jitter}: {fu.describe(p)}\n"); log.flush(); os.fsync(log.fileno())
    try:
        verdict, info = fu.check_device_scene(p, jitter=jitter)
        schedule, policy, clusters = info[:3]
        special += len(info) > 3 and info[3] == 3  # FUZZ_SPECIALISE=1: the scene ran the unit compiled for exactly its types
    except UnsupportedError:  # round 2 refused a sequential fallback batch together with a momentum-conserving angular mode; nothing should be refused any more
        refused += 1
        continue
    split += p["big"] and clusters > 1
    batch_path += clusters == 0
    jittered += jitter != 0 and clusters > 0
    if verdict == "diverged":
        diverged += 1
        continue
    n += 1
    if verdict == "mismatch":
        bad += 1
        print(f"MISMATCH seed {seed} ordinal {ordinal} jitter {jitter}: {fu.describe(p)} schedule/policy/clusters {(schedule, policy, clusters)}", flush=True)
print((f"on units compiled for the scene's exact types {special}; " if os.environ.get("FUZZ_SPECIALISE") == "1" else "") + f"scenes {n} (split-island plans {split}, launch-per-batch {batch_path}, island schedule under jitter {jittered}, refused as UNSUPPORTED {refused}), "
      f"diverged in the oracle and not compared {diverged}, mismatches {bad}")
