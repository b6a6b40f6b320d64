"""Writes the pinning scenes (BEPUPIN1 files) a ReferenceDumper run consumes. Usage: python oracle/pin/export_pin_scenes.py <out_dir>

Scenes: the reference's own microbenchmark inputs (TwoBodyConstraintBenchmarks.cs:42-117), one seeded random-graph scene per supported constraint
type, a mixed-type island scene with a kinematic body, the three scene recipes of BASELINE.json at small sizes, and the floating-point-sensitive cases the C# text
alone does not decide or that sit at the edge of the rational approximations: fast spinners (sin / cos / acos of MathHelper.cs:274-368 far from zero), servos and motors
with MaximumForce = float.MaxValue on every lane, signed zeros (Vector<T> unary minus, wide/wide_vec.h)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import pin_format  # noqa: E402
import small_scenes  # noqa: E402
from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks, SolveDescription  # noqa: E402


def pin_cases():
    """(name, scene, dt, SolveDescription, callbacks, frames)"""
    cb = PoseIntegratorCallbacks()
    for type_id in sorted(TYPE_TABLE):
        yield f"type{type_id:02d}_{TYPE_TABLE[type_id][3]}", small_scenes.random_graph_scene(400 + type_id, 120, 300, [type_id]), 1 / 60, SolveDescription(2, 8), cb, 2
    yield "mixed_islands", small_scenes.island_scene(5, islands=40, bodies_per_island=12, constraints_per_island=40, type_ids=sorted(TYPE_TABLE)), 1 / 60, SolveDescription(1, 4), cb, 3
    yield ("scheduled_iterations", small_scenes.random_graph_scene(7, 200, 500, [0, 7, 22, 25, 30, 47]), 1 / 60,
           SolveDescription(1, 3, velocity_iteration_scheduler=lambda s: [2, 1, 3][s]),
           PoseIntegratorCallbacks(integrate_velocity_for_kinematics=True, allow_substeps_for_unconstrained_bodies=True), 2)
    for mode in (1, 2):
        yield f"angular_mode{mode}", small_scenes.random_graph_scene(60 + mode, 150, 300, [7, 22, 23, 30]), 1 / 60, SolveDescription(1, 4), PoseIntegratorCallbacks(angular_integration_mode=mode), 2
    # ---- floating-point-sensitive cases (round 6) ----
    import numpy as np
    from bepuphysics2_amd.scene import to_aosoa
    # fast spinners: the angular joints with angular velocities of 20-60 rad/s and 2 substeps — half-angles of integrateOrientation up to ~0.25 rad per substep, twist /
    # swing angles swept across the whole range of the rational acos / sin / cos within the three frames
    spin = small_scenes.random_graph_scene(901, 100, 260, [23, 24, 25, 26, 27, 28, 29, 30, 41, 46, 47, 54])
    spin.bodies[:, 12:15] *= np.float32(80.0)
    yield "fp_fast_spinners", spin, 1 / 60, SolveDescription(1, 2), cb, 3
    yield "fp_fast_spinners_conserving", spin.copy(), 1 / 60, SolveDescription(1, 2), PoseIntegratorCallbacks(angular_integration_mode=2), 3
    # MaximumForce / MaximumSpeed = float.MaxValue on EVERY servo and motor lane (the per-type scenes above draw it for half of them): maximumImpulse = MaxValue * dt
    # and the clamps around it (ServoSettings.cs:52-75, MotorSettings.cs)
    forced = small_scenes.random_graph_scene(902, 100, 260, [26, 28, 29, 30, 33, 37, 38, 39, 41, 42, 43, 44, 45, 52, 53])
    fmax = np.float32(np.finfo(np.float32).max)
    for batch in forced.batches:
        for tb in batch:
            lanes = tb.prestep_lanes(forced.bundle_width).copy()
            lanes[(lanes >= 10.0) & (lanes <= 1000.0)] = fmax  # the generator's finite MaximumForce range (synthetic.py: servo(), motor())
            tb.prestep[...] = to_aosoa(lanes, forced.bundle_width)
    yield "fp_max_force_everywhere", forced, 1 / 60, SolveDescription(1, 4), cb, 2
    # signed zeros: bodies at rest with -0.0 velocity components, no gravity, no warm start — every product and negation on the path sees a zero whose sign the C# text
    # fixes only through Vector<T>.op_UnaryNegation (runtime-defined: wide/wide_vec.h)
    zeros = small_scenes.random_graph_scene(903, 80, 200, [0, 3, 4, 7, 22, 23, 25, 30, 31, 47], warm=False)
    zeros.bodies[:, 8:11] = np.float32(-0.0)
    zeros.bodies[:, 12:15] = np.float32(-0.0)
    zeros.bodies[::2, 12:15] = np.float32(0.0)
    yield "fp_signed_zeros", zeros, 1 / 60, SolveDescription(1, 2), PoseIntegratorCallbacks(gravity=(0.0, 0.0, 0.0), linear_damping=0.0, angular_damping=0.0), 1
    try:
        from bepuphysics2_amd.hostlib import HostSimulation
        for name, a in (("pyramid", 3), ("pile", 2000), ("ragdoll_tube", 60)):
            sim = HostSimulation.scene(name, a, 1, 0, 5)
            yield f"recipe_{name}", sim.export(), 1 / 60, sim.solve_description(), cb, 2
            sim.close()
    except Exception as e:  # noqa: BLE001 — the host mirror is optional for exporting
        print("scene recipes skipped:", e)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "pin_scenes"
    os.makedirs(out, exist_ok=True)
    for name, scene, dt, sd, cb, frames in pin_cases():
        pin_format.write_scene(os.path.join(out, name + ".scene.bin"), scene, dt, sd, cb, frames)
        print(name, scene.summary() if hasattr(scene, "summary") else "")


if __name__ == "__main__":
    main()
