#!/usr/bin/env python
"""bench.py — constraint-iterations/s of the MI355X solver + pose-integrator path on the ragdoll-tube scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one Simulation.Solve (prepass + substep loop + final pose integration) over one synthetic scene resident in
HBM. N=1: BASELINE.json configs[2], RagdollTubeBenchmark scaled to ~1M constraints (15,000 ragdolls x (58 joints + 9 synthetic
contacts)), 4 substeps x 1 velocity iteration, dt = 1/60. N>1: configs[3]-style weak scaling — every rank solves its own
independent ragdoll islands of the same size (no cross-GPU contacts, no data-path collective; RCCL barrier only).
Rank 0 prints ONE JSON line; `roofline` and `cpu_baseline` ride on the same line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def build_scene(ragdolls: int, seed: int):
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 0, seed)
    scene = sim.export()
    sd = sim.solve_description()
    sim.close()
    return scene, sd


def cpu_baseline(ragdolls_sample: int, seed: int, target_seconds: float = 12.0):
    """cpu_baseline leg: the oracle's C++ restatement (kind "port"), all host cores, bounded sample of the same workload."""
    import oracle_ffi
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    scene, sd = build_scene(ragdolls_sample, seed)
    cb = PoseIntegratorCallbacks()
    per_frame = scene.constraint_count * int((1 + sd.iterations()).sum())
    # Pick the thread count that is fastest on this host (the barrier-per-batch scheme stops scaling well before 256 threads).
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    candidates = sorted({c for c in (1, 4, 8, 16, 32, 64, 128, avail) if c <= avail})
    best, best_t = 1, float("inf")
    for c in candidates:
        probe = scene.copy()
        t0 = time.perf_counter()
        oracle_ffi.solve(probe, 1 / 60, sd, cb, threads=c, fast=True)
        t = time.perf_counter() - t0
        if t < best_t:
            best, best_t = c, t
        if t > 4 * best_t:
            break
    cores = best
    frames, t0 = 0, time.perf_counter()
    while True:
        oracle_ffi.solve(scene, 1 / 60, sd, cb, threads=cores, fast=True)
        frames += 1
        el = time.perf_counter() - t0
        if el >= target_seconds or frames >= 400:
            break
    return {"value": per_frame * frames / el, "unit": "constraint-iterations/s", "cores": cores, "kind": "port",
            "sample": f"{ragdolls_sample} ragdolls ({scene.constraint_count} constraints), {frames} frames, 4 substeps x 1 iteration, "
                      f"oracle C++ restatement -O3 -march=native, reference work-block/barrier threading, best of thread counts {candidates} on {avail} available CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ragdolls", type=int, default=15000, help="ragdolls per GPU (15000 ~ 1.005M constraints)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")

    from bepuphysics2_amd import build
    build.build_all()  # no-op when the in-tree .so files are current
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.roofline import INTEGRATE_BYTES_PER_BODY, FINAL_BYTES_PER_BODY, scene_stage_bytes
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks

    from bepuphysics2_amd.sharding import rank_seed
    scene, sd = build_scene(args.ragdolls, rank_seed(5, rank))  # each rank: its own independent islands
    cb = PoseIntegratorCallbacks()
    dt = 1.0 / 60.0
    solver = HipSolver(device=local_rank, use_graph=not args.no_graph)
    solver.upload(scene)  # inputs resident in HBM before the timed region starts
    per_step_iterations = scene.constraint_count * int((1 + sd.iterations()).sum())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        solver.sync()

    for _ in range(args.warmup):
        solver.solve(dt, sd, cb, asynchronous=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.solve(dt, sd, cb, asynchronous=True)
    barrier()
    elapsed = time.perf_counter() - t0
    from bepuphysics2_amd import sharding
    units = per_step_iterations * args.steps
    whole_job_rate = sharding.aggregate_throughput(dist, units, elapsed, device=f"cuda:{local_rank}")  # sum(units) / max(elapsed)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(np.isfinite(solver.get_bodies(scene.body_count)).all())

    # ---- roofline of the dominant kernel (batch_kernel<Solve>): instrumented pass, HIP events on the solver's own stream ----
    roofline = None
    if rank == 0:
        ws_bytes, sv_bytes, inc_bytes = scene_stage_bytes(scene)
        solver.set_profiling(True)
        prof_steps = max(3, min(10, args.steps))
        agg = {}
        for _ in range(prof_steps):
            solver.solve(dt, sd, cb)
            for k, (ms, n) in solver.profile().items():
                a = agg.setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += n
        solver.set_profiling(False)
        its = sd.iterations()
        solve_passes = int(its.sum()) * prof_steps
        ws_passes = sd.substep_count * prof_steps
        fam = {}
        for name, total_bytes in (("solve", sv_bytes * solve_passes), ("warmstart", ws_bytes * ws_passes)):
            ms, n = agg[name]
            fam[name] = {"launches": n, "avg_launch_us": 1e3 * ms / max(n, 1), "algorithmic_bytes_per_launch": total_bytes / max(n, 1),
                         "achieved_GBs": total_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}
        dom = "solve" if agg["solve"][0] >= agg["warmstart"][0] else "warmstart"
        roofline = {"bound": "hbm", "kernel": f"batch_kernel<{dom}>", "achieved": fam[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": fam[dom]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_us": fam[dom]["avg_launch_us"], "launches": fam[dom]["launches"],
                    "algorithmic_bytes_per_launch": fam[dom]["algorithmic_bytes_per_launch"],
                    "families_ms_per_step": {k: v[0] / prof_steps for k, v in agg.items()}, "other": fam["warmstart" if dom == "solve" else "solve"],
                    # whole-step view: all algorithmic bytes of a step / step time
                    "step_algorithmic_GBs": (sv_bytes * int(its.sum()) + ws_bytes * sd.substep_count + inc_bytes * (sd.substep_count - 1)
                                             + INTEGRATE_BYTES_PER_BODY * scene.body_count * sd.substep_count + FINAL_BYTES_PER_BODY * scene.body_count)
                                            / (elapsed / args.steps) / 1e9}

    baseline = None
    if rank == 0 and not args.no_cpu_baseline:
        baseline = cpu_baseline(max(args.ragdolls // 8, 64), 5)

    if rank == 0:
        value = whole_job_rate
        out = {
            "metric": "constraint-iterations/sec", "value": value, "unit": "constraint-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RagdollTubeBenchmark scaled to ~1M constraints (BASELINE.json configs[2]): "
                                   f"{args.ragdolls} ragdolls/GPU, {scene.constraint_count} constraints/GPU, {scene.body_count} bodies/GPU, "
                                   f"{len(scene.batches)} batches, {sd.substep_count} substeps x {sd.velocity_iteration_count} velocity iteration(s), dt=1/60",
                       "sharding": "independent ragdoll islands per GPU, no data-path collective" if world > 1 else "single GPU",
                       "hip_graph": not args.no_graph, "finite": finite},
            "roofline": roofline, "cpu_baseline": baseline,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    solver.close()


if __name__ == "__main__":
    main()
