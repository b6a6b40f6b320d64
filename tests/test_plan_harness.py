"""The host-side planner of bepuhip_end_constraints without a device (tools/plan_harness): its output is deterministic, independent of the number of planning threads,
and the experimental cut switches do what they say. The harness includes the library's own translation unit, so this is the product's planner, not a copy."""
import os
import re
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS_DIR = os.path.join(REPO, "tools", "plan_harness")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("plan_harness")
    exe = str(out / "plan_harness")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result", "-Wno-unused-value", "-Wno-array-bounds",
                           "-I", os.path.join(REPO, "include"), "-o", exe, os.path.join(HARNESS_DIR, "plan_harness.hip")])
    scenes = {}
    for kind, size in (("pile", 9000), ("ragdoll_tube", 500), ("graph", 800)):
        path = str(out / f"{kind}.bin")
        subprocess.check_call([sys.executable, os.path.join(HARNESS_DIR, "dump_scene.py"), kind, path, str(size)], stdout=subprocess.DEVNULL)
        scenes[kind] = path
    return exe, scenes


def _run(exe, scene, flags=0, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.check_output([exe, scene, "1", str(flags)], env=e, stderr=subprocess.STDOUT).decode()
    line = [l for l in out.splitlines() if "digest" in l][-1]
    m = re.search(r"enabled (\d) shared (\d) clusters (\d+) items (\d+) .*shared bodies (\d+) \| digest ([0-9a-f]+)", line)
    return {"enabled": int(m.group(1)), "shared": int(m.group(2)), "clusters": int(m.group(3)), "items": int(m.group(4)), "shared_bodies": int(m.group(5)), "digest": m.group(6)}


def test_plan_is_deterministic_and_independent_of_the_planning_threads(harness):
    exe, scenes = harness
    for kind, path in scenes.items():
        for flags in (0, 8):  # 8 = BEPUHIP_FLAG_RESERVE_UPDATE_SLOTS
            runs = [_run(exe, path, flags, BEPUHIP_PLAN_THREADS=threads) for threads in (1, 3, 8, 8)]
            assert len({r["digest"] for r in runs}) == 1, (kind, flags, runs)
            assert runs[0]["enabled"] == 1, (kind, flags, runs[0])
    assert _run(exe, scenes["ragdoll_tube"], 0)["digest"] != _run(exe, scenes["ragdoll_tube"], 8)["digest"]  # reserved slots change a whole-island layout ...
    assert _run(exe, scenes["pile"], 0)["digest"] != _run(exe, scenes["pile"], 8)["digest"]  # ... and, since round 3, a split-island one (free row slots per cluster segment, free LDS slots per cluster)


def test_one_connected_pile_is_cut_and_the_cut_switches_reduce_the_shared_bodies(harness):
    exe, scenes = harness
    first_body = _run(exe, scenes["pile"], BEPUHIP_SPLIT_CLUSTERS=24, BEPUHIP_SPLIT_COVER=0, BEPUHIP_SPLIT_REFINE=0)  # round 2's rule: the first dynamic body's cluster runs a crossing constraint
    assert first_body["shared"] == 1 and first_body["clusters"] > 1 and first_body["shared_bodies"] > 0
    cover = _run(exe, scenes["pile"], BEPUHIP_SPLIT_CLUSTERS=24, BEPUHIP_SPLIT_COVER=1, BEPUHIP_SPLIT_REFINE=0)
    refined = _run(exe, scenes["pile"], BEPUHIP_SPLIT_CLUSTERS=24, BEPUHIP_SPLIT_COVER=1, BEPUHIP_SPLIT_REFINE=2)
    assert cover["enabled"] == refined["enabled"] == 1
    assert refined["shared_bodies"] <= cover["shared_bodies"] < first_body["shared_bodies"]
    assert _run(exe, scenes["pile"], BEPUHIP_SPLIT_CLUSTERS=24)["digest"] == refined["digest"]  # the default plan since round 3: vertex cover + two refine sweeps
    assert _run(exe, scenes["pile"], BEPUHIP_NO_SPLIT=1)["enabled"] == 0  # the launch-per-batch schedule takes the scene
    whole = _run(exe, scenes["ragdoll_tube"])
    assert whole["shared"] == 0 and whole["shared_bodies"] == 0  # islands that fit are never cut


def _validate(exe, scene, flags=0, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    e["PLAN_VALIDATE"] = "1"
    r = subprocess.run([exe, scene, "1", str(flags)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert "validate:" in out, out  # the plan was enabled and checked
    return r.returncode, out


def test_plans_keep_the_invariants_the_device_relies_on(harness):
    """Checked from the plan's output alone: every live row in exactly one item of its batch, local references and slot tables agree with the global references, a body at most
    once per batch and cluster, predecessors from earlier batches, and for split plans one home per shared body, ghost slots elsewhere, ranks 0..d-1 in batch order with the
    right degree — for the default plans, the reserved-slot layout and the experimental cuts. Three deliberate corruptions are noticed."""
    exe, scenes = harness
    for kind, path in scenes.items():
        for flags in (0, 8):
            rc, out = _validate(exe, path, flags)
            assert rc == 0, (kind, flags, out)
    for env in (dict(BEPUHIP_SPLIT_CLUSTERS=24), dict(BEPUHIP_SPLIT_CLUSTERS=24, BEPUHIP_SPLIT_COVER=1), dict(BEPUHIP_SPLIT_CLUSTERS=40, BEPUHIP_SPLIT_COVER=1, BEPUHIP_SPLIT_REFINE=3),
                dict(BEPUHIP_SPLIT_CLUSTERS=24, BEPUHIP_SPLIT_SEPARATE=1)):
        rc, out = _validate(exe, scenes["pile"], **env)
        assert rc == 0, (env, out)
    for mutation in (1, 2, 3):
        rc, out = _validate(exe, scenes["pile"], BEPUHIP_SPLIT_CLUSTERS=24, PLAN_MUTATE=mutation)
        assert rc == 3 and "plan violation" in out, (mutation, out)


def test_structural_churn_keeps_a_split_plan_valid_and_its_device_image_equal_to_the_host_mirrors(harness):
    """PLAN_CHURN (tools/plan_harness): the library's own add / remove code runs on the plan's host mirrors, frame after frame; after every frame the rows a device would hold
    — the upload, then nothing but what the flush sends: whole slots and single words — equal the mirrors, and the mirrors still pass every plan invariant. Near pairs (what a
    narrow phase produces) stay on the plan; pairs from all over the scene use the free LDS slots up and are REFUSED cleanly (the context then leaves the plan)."""
    exe, scenes = harness
    e = dict(os.environ, PLAN_VALIDATE="1", PLAN_CHURN="25", BEPUHIP_SPLIT_CLUSTERS="24")
    r = subprocess.run([exe, scenes["pile"], "1", "8"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "25 frames" in out and "still on the plan (24 clusters)" in out, out
    assert out.count("validate: 0 violation(s)") == 26, out  # the plan, then every frame
    # ... and with bodies that leave the plan (all their constraints removed), bodies that move to another index (Bodies.RemoveAt: every reference patched) and bodies that
    # join (a free index gets a constraint): soft_release_body / soft_move_body / soft_adopt_body on a split plan, shared bodies and their ghost copies included
    e["PLAN_CHURN_BODIES"] = "1"
    r = subprocess.run([exe, scenes["pile"], "1", "8"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "still on the plan (24 clusters)" in out and "25 bodies moved to another index, 25 bodies joined" in out, out
    assert out.count("validate: 0 violation(s)") == 26, out
    # the same on a whole-island plan (islands packed into clusters, no shared bodies): new pairs inside an island, bodies leaving / moving / joining
    r = subprocess.run([exe, scenes["ragdoll_tube"], "1", "8"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "25 frames" in out and "still on the plan" in out and "25 bodies moved to another index, 25 bodies joined" in out, out
    assert out.count("validate: 0 violation(s)") == 26, out
    e.pop("PLAN_CHURN_BODIES")
    e["PLAN_CHURN_FAR"] = "1"
    r = subprocess.run([exe, scenes["pile"], "1", "8"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "violation(s)" in out and "difference(s)" not in out, out  # valid until the refusal, if there is one
    e.pop("PLAN_CHURN_FAR")
    r = subprocess.run([exe, scenes["pile"], "1", "0"], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)  # without reserved slots: the first new pair finds no room
    assert r.returncode == 0 and "difference(s)" not in r.stdout.decode(), r.stdout.decode()
