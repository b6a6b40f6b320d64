"""Same-box A/B timing of library switches on one of the bench scenes. Not part of the product.

    python tools/ab_scene.py ragdoll|pile|crowd "name:VAR=value,VAR=value" ...

Every configuration gets a fresh context (the switches are read at upload / solve time), the same uploaded scene, 100 warm-up solves, then two timed runs of STEPS
solves; the bodies after exactly the same number of steps must be bit-identical across configurations (every switch here is a schedule, never a result)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import PoseIntegratorCallbacks  # noqa: E402

SCENES = {"ragdoll": ("ragdoll_tube", int(os.environ.get("RAGDOLLS", "15000")), 1, 0, 5), "pile": ("pile", int(os.environ.get("BOXES", "100000")), 0, 0, 5),
          "crowd": ("ragdoll_tube", int(os.environ.get("RAGDOLLS", "15000")), 1, 2, 5)}


def main():
    name = sys.argv[1]
    configs = sys.argv[2:] or ["default:"]
    sim = HostSimulation.scene(*SCENES[name])
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    cb = PoseIntegratorCallbacks()
    its = scene.constraint_count * int((1 + sd.iterations()).sum())
    steps = int(os.environ.get("STEPS", "200"))
    print(f"{name}: {scene.body_count} bodies, {scene.constraint_count} constraints, {len(scene.batches)} batches, {sd.substep_count} substeps x {list(map(int, sd.iterations()))}", flush=True)
    reference = None
    for spec in configs:
        label, _, assignments = spec.partition(":")
        changed = {}
        for a in filter(None, assignments.split(",")):
            k, _, v = a.partition("=")
            changed[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            s = HipSolver()
            s.upload(scene)
            if os.environ.get("AB_SPECIALISE"):  # the unit compiled for exactly the scene's types (and this configuration's workgroup size), waited for
                print(f"  ({label}: specialise_units -> {s.specialise_units(wait=True)})", flush=True)
            for _ in range(100):
                s.solve(1 / 60, sd, cb, asynchronous=True)
            s.sync()
            times = []
            for _ in range(2):
                t0 = time.perf_counter()
                for _ in range(steps):
                    s.solve(1 / 60, sd, cb, asynchronous=True)
                s.sync()
                times.append((time.perf_counter() - t0) / steps * 1e3)
            bodies = s.get_bodies(scene.body_count)
            cyc = s.cluster_cycles()
            s.close()
            same = "reference" if reference is None else ("bit-identical" if np.array_equal(reference.view(np.int32), bodies.view(np.int32)) else "DIFFERENT RESULT")
            if reference is None:
                reference = bodies
            print(f"  {label:<28s} {min(times):.4f} ms/step (runs {', '.join(f'{t:.4f}' for t in times)})  {its / min(times) / 1e6:.2f} G c-it/s  clusters {cyc.size}"
                  f"  kcycles max {float(cyc.max()) / 1e3 if cyc.size else 0:.0f}  {same}  finite {bool(np.isfinite(bodies).all())}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"  {label:<28s} FAILED: {e}", flush=True)
        finally:
            for k, v in changed.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


if __name__ == "__main__":
    main()
