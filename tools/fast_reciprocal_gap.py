"""How far is the portable branch of MathHelper.FastReciprocal[SquareRoot] (`1 / v`, `1 / sqrt(v)`: what the device and both oracles compute) from the branch the reference
takes on an AVX host (vrcpps / vrsqrtps, BepuUtilities/MathHelper.cs:384,401)? CPU only: oracle/wide built twice (Makefile: libbepu_wide.so, libbepu_wide_rcpx86.so), the
same scenes through both, relative velocity error = max over bodies of |v_portable - v_x86| / max |v_x86| (BASELINE.json's measure). Per type that uses the helpers
(CenterDistanceConstraint 35, CenterDistanceLimit 55, AreaConstraint 36, VolumeConstraint 32) and for a mixed scene, after 8 substeps (north_star's horizon) and after
60 frames x 4 substeps — next to what a one-off perturbation of the initial velocities by 1e-4 (relative, the tolerance itself) grows into over the same 60 frames (the
scenes are random constraint graphs: chaotic), so that the long-horizon figure can be read for what it is. The vrcpps / vrsqrtps results are those of THIS host's CPU (the low bits differ between vendors).
    python tools/fast_reciprocal_gap.py [seeds]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import small_scenes
import wide_ffi
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

VEL = [8, 9, 10, 12, 13, 14]
CASES = (("CenterDistanceConstraint", [35]), ("CenterDistanceLimit", [55]), ("AreaConstraint", [36]), ("VolumeConstraint", [32]),
         ("mixed (the four + BallSocket, Contact4, Contact1)", [35, 55, 36, 32, 22, 7, 4]))


def run(scene, variant, frames, substeps):
    s, sd, cb = scene.copy(), SolveDescription(1, substeps), PoseIntegratorCallbacks()
    for _ in range(frames):
        wide_ffi.solve(s, 1 / 60, sd, cb, variant=variant)
    return s


def relative_velocity_error(ref, got):
    return float(np.abs(ref.bodies[:, VEL] - got.bodies[:, VEL]).max() / max(float(np.abs(ref.bodies[:, VEL]).max()), 1e-6))


def gap(types, seed, frames, substeps):
    scene = small_scenes.random_graph_scene(seed, 300, 600, types)
    x86 = run(scene, "rcpx86", frames, substeps)
    return relative_velocity_error(x86, run(scene, "", frames, substeps)), scene, x86


def tolerance_sized_nudge_growth(scene, frames, substeps, seed):
    """What a ONE-OFF relative perturbation of 1e-4 (the tolerance itself) of every velocity component grows into."""
    nudged = scene.copy()
    rng = np.random.default_rng(seed)
    nudged.bodies[:, VEL] *= (1.0 + 1e-4 * rng.uniform(-1, 1, nudged.bodies[:, VEL].shape)).astype(np.float32)
    return relative_velocity_error(run(scene, "", frames, substeps), run(nudged, "", frames, substeps))


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cpu = next((line.split(":", 1)[1].strip() for line in open("/proc/cpuinfo") if line.startswith("model name")), "?")
    print(f"host CPU: {cpu}; {seeds} random graphs (300 bodies, 600 constraints) per row; relative velocity error of the portable branch against vrcpps / vrsqrtps")
    print(f"{'types':52s} {'8 substeps: max':>16s} {'median':>10s} | {'60 frames x 4: max':>19s} | {'1e-4 nudge of v0 after 60 frames: max':>38s}")
    for name, types in CASES:
        short = [gap(types, seed, 1, 8)[0] for seed in range(1, seeds + 1)]
        long_ = [gap(types, seed, 60, 4)[0] for seed in range(1, seeds + 1)]
        ulp = [tolerance_sized_nudge_growth(small_scenes.random_graph_scene(seed, 300, 600, types), 60, 4, seed) for seed in range(1, seeds + 1)]
        print(f"{name:52s} {max(short):16.3e} {float(np.median(short)):10.3e} | {max(long_):19.3e} | {max(ulp):38.3e}")
