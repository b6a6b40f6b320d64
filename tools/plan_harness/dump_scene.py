"""Writes a scene in the plan harness's binary format:  python tools/plan_harness/dump_scene.py <pile|ragdoll_tube|crowd|graph|graph44|fuzz:seed:ordinal> <out.bin> [size]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np

kind, out = sys.argv[1], sys.argv[2]
size = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if kind.startswith("fuzz:"):  # fuzz:<seed>:<ordinal> — the scene of a tools/fuzz_device.py ordinal
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import replay_fuzz_device as rf
    import small_scenes
    _, fuzz_seed, ordinal = kind.split(":")
    p = rf.parameters(int(fuzz_seed), int(ordinal) + 1)[int(ordinal)]
    scene = small_scenes.random_graph_scene(p["seed"], p["nb"], p["nc"], p["types"], kinematic_fraction=p["kin"])
elif kind == "graph44":  # every constraint type id, three- and four-body ones included
    import small_scenes
    scene = small_scenes.random_graph_scene(7, size or 6000, (size or 6000) * 2, sorted(small_scenes.TYPE_TABLE))
elif kind == "graph":
    import small_scenes
    scene = small_scenes.random_graph_scene(7, size or 6000, (size or 6000) * 2, sorted(t for t, i in small_scenes.TYPE_TABLE.items() if i[0] <= 2))
else:
    from bepuphysics2_amd.hostlib import HostSimulation
    args = {"pile": ("pile", size or 100000, 0, 0, 5), "ragdoll_tube": ("ragdoll_tube", size or 15000, 1, 0, 1), "crowd": ("ragdoll_tube", size or 15000, 1, 2, 11)}[kind]
    scene = HostSimulation.scene(*args).export()
tbs = [(bi, tb) for bi, b in enumerate(scene.batches) for tb in b]
with open(out, "wb") as f:
    np.asarray([scene.bundle_width, len(scene.batches), len(tbs), 64], np.int32).tofile(f)
    for bi, tb in tbs:
        np.asarray([bi, tb.type_id, tb.count, tb.body_refs.size, tb.prestep.size, tb.accumulated.size], np.int32).tofile(f)
        tb.body_refs.astype(np.int32).tofile(f); tb.prestep.astype(np.float32).tofile(f); tb.accumulated.astype(np.float32).tofile(f)
print(f"{kind}: {scene.body_count} bodies, {scene.constraint_count} constraints, {len(scene.batches)} batches -> {out}")
