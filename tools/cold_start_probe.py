"""Why do the first tens of ASYNCHRONOUS steps of a connected scene run at a third of the settled rate (round 5: bench.py --lattice --lattice-exact 0.78 ms/step over its
first 60 steps, 0.26 after 300 pre-warm steps; round 5's probe, one solve + one sync at a time, saw nothing)? Round 6: the steps are enqueued back to back in chunks of
ten as the bench does, every chunk timed, and the device's clock levels (sysfs pp_dpm_sclk / mclk / fclk / socclk, the line marked '*') read between chunks —
  A  connected lattice, cold device (nothing ran for two seconds)
  B  the same after 60 ms of a bandwidth-bound load (device-to-device copies of 256 MB)
  C  the same after two idle seconds again (does it fall back?)
  D  the headline scene (whole islands: no hand-offs between clusters), cold, for comparison.
Developer probe, not part of the product."""
import glob
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bepuphysics2_amd.hostlib import HostSimulation  # noqa: E402
from bepuphysics2_amd.native import HipSolver  # noqa: E402
from bepuphysics2_amd.scene import PoseIntegratorCallbacks  # noqa: E402


def clocks():
    out = []
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        level = "?"
        for path in glob.glob(f"/sys/class/drm/card*/device/{name}"):
            try:
                for line in open(path):
                    if "*" in line:
                        level = line.split(":")[1].replace("*", "").strip()
            except OSError:
                pass
        out.append(f"{name[7:]} {level}")
    return ", ".join(out)


def chunks(solver, sd, cb, n_chunks, label):
    times = []
    for k in range(n_chunks):
        t0 = time.perf_counter()
        for _ in range(10):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.sync()
        times.append(1e2 * (time.perf_counter() - t0))
        if k in (0, 2, 5, 11, n_chunks - 1):
            print(f"    after {10 * (k + 1):4d} steps: {clocks()}", flush=True)
    print(f"  {label}: ms/step per chunk of ten: " + " ".join(f"{t:.3f}" for t in times), flush=True)


def scene_of(*args):
    sim = HostSimulation.scene(*args)
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    return scene, sd


import torch  # noqa: E402
cb = PoseIntegratorCallbacks()
lattice, sd = scene_of("ragdoll_tube", 15000, 1, 1, 5)


# ---- E: bench.py --lattice --lattice-exact --no-prewarm, step for step: a fresh context, upload, 10 + 50 asynchronous steps with no pause anywhere; variants ----
def bench_like(label, group, pause, chunked):
    s = HipSolver(exclusive_device=True)
    if group:
        s.set_device_group(1, 0)
    s.upload(lattice, sd.fallback_batch_threshold)
    if pause:
        time.sleep(pause)
    out, slow_calls = [], []
    for steps in (10, 50, 50, 50, 50, 50):
        t0 = time.perf_counter()
        for k in range(steps):
            c0 = time.perf_counter()
            s.solve(1 / 60, sd, cb, asynchronous=True)
            c1 = time.perf_counter()
            if c1 - c0 > 1e-3:
                import gc
                slow_calls.append(f"solve call {k} of a loop of {steps}: {1e3 * (c1 - c0):.1f} ms on the host (gc counts {gc.get_count()}, full collections so far {gc.get_stats()[2]['collections']})")
            if group:
                s.sync_owned_bodies()
            if chunked and k % 10 == 9:
                s.sync()
        c0 = time.perf_counter()
        s.sync()
        if time.perf_counter() - c0 > 0.02:
            slow_calls.append(f"sync after a loop of {steps}: {1e3 * (time.perf_counter() - c0):.1f} ms")
        out.append(1e3 * (time.perf_counter() - t0) / steps)
    s.close()
    print(f"  E {label}: ms/step over 10 + 5 x 50 steps: " + " ".join(f"{t:.3f}" for t in out) + (" | " + "; ".join(slow_calls) if slow_calls else ""), flush=True)



if len(sys.argv) > 5 and sys.argv[5] == "nogc":  # the suspect: a full collection of Python's cyclic garbage collector (tens of milliseconds with torch imported) landing in a timed loop
    import gc
    gc.collect()
    gc.disable()
if len(sys.argv) > 1:  # `fresh <group> <pause> <chunked> [nogc]`: the bench-like sequence as the FIRST device work of a fresh process
    bench_like(f"first work of a fresh process (group {sys.argv[2]}, pause {sys.argv[3]}, sync every ten {sys.argv[4]})", sys.argv[2] == "1", float(sys.argv[3]), sys.argv[4] == "1")
    bench_like("the same process, second context", sys.argv[2] == "1", float(sys.argv[3]), sys.argv[4] == "1")
    sys.exit(0)
solver = HipSolver(exclusive_device=True)
solver.upload(lattice)
print("idle:", clocks(), flush=True)
time.sleep(2.0)
chunks(solver, sd, cb, 40, "A lattice, cold")
time.sleep(2.0)
a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.06:
    b.copy_(a)
torch.cuda.synchronize()
print("  after 60 ms of copies:", clocks(), flush=True)
chunks(solver, sd, cb, 12, "B lattice right after a bandwidth-bound load")
time.sleep(2.0)
chunks(solver, sd, cb, 12, "C lattice after two idle seconds")
solver.close()
headline, sd2 = scene_of("ragdoll_tube", 15000, 1, 0, 5)
solver = HipSolver()
solver.upload(headline)
time.sleep(2.0)
chunks(solver, sd2, cb, 12, "D headline scene (whole islands), cold")
solver.close()


bench_like("as the bench does (device group of one, no pause)", True, 0.0, False)
bench_like("without set_device_group", False, 0.0, False)
bench_like("group of one, 2 s pause after the upload", True, 2.0, False)
bench_like("group of one, a sync every ten steps", True, 0.0, True)
bench_like("as the bench does, again", True, 0.0, False)
