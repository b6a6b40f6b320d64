"""The oracle's pinning kit (oracle/pin): exchange format round trip and the comparison, with the oracle standing in for the C# ReferenceDumper."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle", "pin"))

import oracle_ffi
import small_scenes
from bepuphysics2_amd.scene import PoseIntegratorCallbacks, SolveDescription


def test_scene_export_and_result_comparison_round_trip(tmp_path):
    import compare_with_reference as cmp
    import pin_format
    scene = small_scenes.random_graph_scene(9, 60, 150, [3, 7, 10, 17, 22, 31, 47])
    sd = SolveDescription(1, 3, velocity_iteration_scheduler=lambda s: [2, 1, 3][s])
    cb = PoseIntegratorCallbacks(gravity=(0.5, -9.0, 0.25), integrate_velocity_for_kinematics=True, angular_integration_mode=1)
    path = str(tmp_path / "case.scene.bin")
    pin_format.write_scene(path, scene, 1 / 60, sd, cb, frames=2)
    data = open(path, "rb").read()
    assert data[:8] == b"BEPUPIN1" and int.from_bytes(data[8:12], "little") == scene.body_count
    # the scene file carries every float of the scene: bodies + 9 header words + per-constraint lanes
    lanes = sum(tb.count * (tb.bodies + tb.prestep_floats + tb.impulse_floats) for b in scene.batches for tb in b)
    tbs = sum(len(b) for b in scene.batches)
    assert len(data) == 8 + 4 + scene.body_count * 128 + 8 * 4 + (3 + 3) * 4 + 2 * 4 + 4 + len(scene.batches) * 4 + tbs * 20 + lanes * 4
    # a "reference" result produced by the oracle itself compares bit-exact ...
    advanced = scene.copy()
    for _ in range(2):
        oracle_ffi.solve(advanced, 1 / 60, sd, cb)
    result = str(tmp_path / "case.result.bin")
    pin_format.write_result(result, advanced)
    m = cmp.compare_case(scene, 1 / 60, sd, cb, 2, result)
    assert m["bodies_bit_exact"] and m["impulses_bit_exact"] and m["prestep_bit_exact"] and m["velocity_rel_err"] == 0.0
    # ... and a single flipped mantissa bit in one velocity is reported
    advanced.bodies[5, 9] = np.nextafter(advanced.bodies[5, 9], np.float32(np.inf))
    pin_format.write_result(result, advanced)
    m = cmp.compare_case(scene, 1 / 60, sd, cb, 2, result)
    assert not m["bodies_bit_exact"] and m["bodies_max_ulp"] == 1


def test_every_supported_type_is_exported():
    from export_pin_scenes import pin_cases
    from bepuphysics2_amd.scene import TYPE_TABLE
    names = [c[0] for c in pin_cases()]
    for type_id, row in TYPE_TABLE.items():
        assert f"type{type_id:02d}_{row[3]}" in names
    assert len(names) >= len(TYPE_TABLE) + 4
