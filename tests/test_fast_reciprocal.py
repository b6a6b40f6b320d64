"""The one parity gap that can be measured without the reference: MathHelper.FastReciprocal[SquareRoot] (BepuUtilities/MathHelper.cs:380-412). The device and both oracles
restate its portable branch (`1 / v`, `1 / sqrt(v)`); on an AVX host — the CPU beside the GPU — the reference takes `Avx.Reciprocal` / `Avx.ReciprocalSqrt`, i.e. vrcpps /
vrsqrtps (relative error <= 1.5 * 2^-12, low bits vendor-specific). oracle/wide built with -DWIDE_FAST_RECIPROCAL_X86 uses those very instructions; this test pins how
far the portable branch is from it after north_star's 8 substeps, for the four types that call the helpers (CenterDistanceConstraint.cs:87,103, CenterDistanceLimit.cs:86,
AreaConstraint.cs:137, VolumeConstraint.cs:122). Table and long horizon: tools/fast_reciprocal_gap.py -> profiles/r06_fast_reciprocal_gap.txt, DESIGN.md §4. CPU only."""
import numpy as np
import pytest

import small_scenes
import wide_ffi
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription

VEL = [8, 9, 10, 12, 13, 14]
INSTRUCTION_BOUND = 1.5 * 2.0 ** -12  # Intel SDM / AMD APM: |relative error| of (V)RCPPS and (V)RSQRTPS


def _solve(scene, variant, substeps=8):
    s = scene.copy()
    wide_ffi.solve(s, 1 / 60, SolveDescription(1, substeps), PoseIntegratorCallbacks(), variant=variant)
    return s


def _gap(types, seed):
    scene = small_scenes.random_graph_scene(seed, 300, 600, types)
    x86, portable = _solve(scene, "rcpx86"), _solve(scene, "")
    assert np.isfinite(x86.bodies[:, :15]).all() and np.isfinite(portable.bodies[:, :15]).all()
    return float(np.abs(x86.bodies[:, VEL] - portable.bodies[:, VEL]).max() / max(float(np.abs(x86.bodies[:, VEL]).max()), 1e-6))


# (type ids, the bound this repository states for "relative velocity error after 8 substeps against an x86 reference")
@pytest.mark.parametrize("name,types,stated", [("CenterDistanceConstraint", [35], 4e-4), ("CenterDistanceLimit", [55], 4e-4), ("AreaConstraint", [36], 1e-4),
                                               ("VolumeConstraint", [32], 1e-4), ("mixed", [35, 55, 36, 32, 22, 7, 4], 4e-4)])
def test_portable_branch_against_the_x86_instructions_after_eight_substeps(name, types, stated):
    gaps = [_gap(types, seed) for seed in (1, 2, 3)]
    # the two branches DO differ (otherwise the variant is not what it says) ...
    assert min(gaps) > 0.0, (name, gaps)
    # ... by no more than the stated bound: the centre-distance types sit AT north_star's 1e-4 (0.6e-4 .. 1.3e-4 measured on Intel and AMD hosts: above it on some
    # scenes), the area / volume types an order of magnitude below. The instruction's own error bound is the ceiling of what one evaluation can contribute.
    assert max(gaps) <= stated, (name, gaps)
    assert stated <= INSTRUCTION_BOUND * 1.1


def test_the_variant_changes_nothing_for_scenes_without_the_four_types():
    scene = small_scenes.random_graph_scene(7, 200, 500, [22, 23, 25, 7, 4, 0, 30, 47, 24])
    a, b = _solve(scene, "rcpx86", 4), _solve(scene, "", 4)
    assert np.array_equal(a.bodies.view(np.int32), b.bodies.view(np.int32))
