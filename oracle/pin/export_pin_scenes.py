"""Writes the pinning scenes (BEPUPIN1 files) a ReferenceDumper run consumes. Usage: python oracle/pin/export_pin_scenes.py <out_dir>

Scenes: the reference's own microbenchmark inputs (TwoBodyConstraintBenchmarks.cs:42-117), one seeded random-graph scene per supported constraint
type, a mixed-type island scene with a kinematic body, and the three scene recipes of BASELINE.json at small sizes."""
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
