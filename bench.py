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
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def tests_on_path():
    """tests/ holds the checker's bindings (wide_ffi: the cpu_baseline leg, the only place bench.py may touch oracle/) and the scene generator of the all-44 leg: put on
    sys.path by the two legs that need it, not at import."""
    path = os.path.join(REPO, "tests")
    if path not in sys.path:
        sys.path.insert(0, path)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def cpu_quota():
    """The CPU time this process may use per period, from the cgroup (v2 cpu.max, v1 cpu.cfs_quota_us): a container with 256 visible CPUs and a quota of 8 runs at most
    8 threads' worth of work however many it starts. Returns (cpus or None when unlimited, where it was read)."""
    for path, v2 in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False), ("/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_quota_us", False)):
        try:
            text = open(path).read().split()
            if v2:
                return (None if text[0] == "max" else float(text[0]) / float(text[1])), path
            quota = float(text[0])
            period = float(open(path.replace("cfs_quota_us", "cfs_period_us")).read().split()[0])
            return (None if quota <= 0 else quota / period), path
        except (OSError, ValueError, IndexError):
            continue
    return None, None


def specialise(solver):
    """Setup, untimed: the island kernel compiled for exactly the scene's constraint types (bepuhip_specialise_units; the units of the BASELINE.json scenes are prebuilt by
    __graft_entry__.build() into csrc/units/ and load in milliseconds, anything else is a 40-60 s hipcc run on a thread of the library — waited for here, like any JIT
    warm-up). BENCH_SPECIALISE=0 keeps the prebuilt families. Returns the state (2 = loaded)."""
    if os.environ.get("BENCH_SPECIALISE", "1") == "0":
        return -1
    return solver.specialise_units(wait=True)


def compact_line(out, full_path):
    """The ONE line the driver keeps (its record holds an 8 KB tail): every number, none of the prose. The long form — notes, methods, thread curves, per-size
    details — goes to `full_path`. Headline keys first; pile and crowd ride inside `roofline` as well so that a cut tail still shows them."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}

    def r4(x):
        return float(f"{x:.4g}") if isinstance(x, float) else x

    def rnd(d):
        if isinstance(d, dict):
            return {k: rnd(v) for k, v in d.items()}
        if isinstance(d, list):
            return [rnd(v) for v in d]
        return r4(d)

    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = dict(out["config"])
    if cfg.get("row_policy"):
        cfg["row_policy"] = cfg["row_policy"].split(",")[0]
    line["config"] = cfg
    roof = out.get("roofline")
    if isinstance(roof, dict):
        keep = pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "basis", "avg_launch_us", "launches", "algorithmic_bytes_per_launch", "algorithmic_GBs",
                           "algorithmic_frac_of_peak", "memory_stream_bytes_per_launch", "memory_stream_GBs", "traffic_over_compulsory_stream", "working_set_bytes", "valu_busy",
                           "effective_shader_GHz", "step_algorithmic_GBs"))
        issue = (roof.get("traffic_detail") or {}).get("issue") if isinstance(roof.get("traffic_detail"), dict) else None
        if issue:
            keep["waves_per_simd"], keep["wave_time_waiting"] = issue.get("waves_per_simd"), issue.get("wave_time_waiting")
        for short, key in (("pile", "pile_100k"), ("crowd", "ragdoll_crowd")):
            leg = (out.get("connected_scenes") or {}).get(key)
            if isinstance(leg, dict) and "ms_per_step" in leg:
                lr = leg.get("roofline") or {}
                keep[f"{short}_ms"], keep[f"{short}_frac"] = leg["ms_per_step"], lr.get("frac")
                keep[f"{short}_traffic_over_stream"], keep[f"{short}_valu_busy"] = lr.get("traffic_over_compulsory_stream"), lr.get("valu_busy")
        line["roofline"] = rnd(keep)
    else:
        line["roofline"] = roof
    base = out.get("cpu_baseline")
    if isinstance(base, dict) and "value" in base:
        keep = pick(base, ("value", "unit", "cores", "kind", "single_thread_value", "host_cpus_available", "cpu_quota", "gpu_over_cpu", "gpu_over_ideal_socket"))
        keep["ideal_socket_bound"] = (base.get("ideal_socket_bound") or {}).get("value")
        keep["thread_curve"] = [[e["threads"], r4(float(e["value"]))] for e in base.get("thread_curve", [])]
        keep["sample"] = base.get("sample", "")[:160]
        line["cpu_baseline"] = rnd(keep)
    else:
        line["cpu_baseline"] = base
    conn = out.get("connected_scenes")
    if isinstance(conn, dict):
        line["connected_scenes"] = {}
        for key, leg in conn.items():
            if isinstance(leg, dict) and "ms_per_step" in leg:
                lr = leg.get("roofline") or {}
                line["connected_scenes"][key] = rnd(dict(pick(leg, ("ms_per_step", "value", "upload_ms", "finite")), frac=lr.get("frac"), traffic=lr.get("traffic"),
                                                       traffic_over_compulsory_stream=lr.get("traffic_over_compulsory_stream"), valu_busy=lr.get("valu_busy"),
                                                       island_schedule=str(leg.get("schedule", "")).startswith("island")))
            else:
                line["connected_scenes"][key] = leg
    sweep = out.get("scale_sweep")
    if isinstance(sweep, dict) and "sizes" in sweep:
        line["scale_sweep"] = rnd([pick(e, ("constraints", "ms_per_step", "value", "frac", "working_set_bytes", "clusters_per_cu")) for e in sweep["sizes"]])
    elif sweep is not None:
        line["scale_sweep"] = sweep
    wid = out.get("widened_types")
    line["widened_types"] = rnd(pick(wid, ("ms_per_step", "value", "finite", "error", "all_types_ms", "all_types_value", "all_types_finite", "all_types_error"))) if isinstance(wid, dict) else wid
    bnd = out.get("boundary")
    line["boundary"] = rnd({k: v for k, v in bnd.items() if not isinstance(v, str) or k == "error"}) if isinstance(bnd, dict) else bnd
    lat = out.get("lattice")
    if isinstance(lat, dict):
        line["lattice"] = rnd({k: (pick(v, ("within_north_star_tolerance", "velocity_err_max", "bit_identical", "exchanges_per_frame", "schedules")) if isinstance(v, dict) else v)
                               for k, v in lat.items() if k in ("per_pass_block_jacobi", "per_batch_exact", "device_group_exact", "error")})
    for extra in ("all_types", "frame"):
        if extra in out:
            line[extra] = rnd(out[extra])
    grp = out.get("lattice_device_group")  # N > 1: BASELINE.json configs[4] (one connected lattice over the N devices, exact) beside configs[3]
    if isinstance(grp, dict):
        line["lattice_device_group"] = rnd(dict(pick(grp, ("value", "ms_per_step", "scaling", "n_gpus", "error", "ranks")), **pick(grp.get("config") or {}, ("exchanges_per_step", "finite", "schedule_is_island"))))
    if out.get("self_checks"):
        line["self_checks"] = out["self_checks"]
    line["full_report"] = full_path
    return line


def build_scene(ragdolls: int, seed: int):
    from bepuphysics2_amd.hostlib import HostSimulation
    sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 0, seed)
    scene = sim.export()
    sd = sim.solve_description()
    sim.close()
    return scene, sd


def cpu_baseline(ragdolls_sample: int, seed: int, target_seconds: float = 14.0):
    """cpu_baseline leg: oracle/wide — the reference's CPU path restated in its own shape (AOSOA bundles of 8 lanes, AVX2 8x8
    transposed gather/scatter, fused first-touch integration, the work-block / claim / sync-stage scheduler of Solver_Solve.cs:297-946) — on the
    host cores, bounded sample of the same workload. kind "port-simd8": a C++ port, not the RyuJIT binary (no .NET in the image).

    The scene lives in a persistent session (oracle/wide/wide_solver.cpp `Session`): 128-byte aligned buffers owned by the library, the batches' handle sets built
    once — as the reference holds a simulation between frames (BufferPool.cs:42, Solver.cs:1046-1051). The timed region is exactly Simulation.Solve
    (Simulation.cs:278-290): PrepareConstraintIntegrationResponsibilities + Solve + IntegrateAfterSubstepping. No marshalling, no copies, no rebuilds inside it."""
    tests_on_path()
    import wide_ffi
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    scene, sd = build_scene(ragdolls_sample, seed)
    cb = PoseIntegratorCallbacks()
    per_frame = scene.constraint_count * int((1 + sd.iterations()).sum())
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    session = wide_ffi.Session(scene, 1 / 60, sd, cb, fast=True)
    pin = wide_ffi.pin_plan("fast")
    curve = []
    budget_each = target_seconds / 9.0
    quota, quota_path = cpu_quota()
    counts = {1, 8, 16, 32, 64, 128, pin["first_socket_physical_cores"]}  # ... and exactly one worker per physical core of the socket
    if quota:  # ... and exactly the cgroup's CPU quota, when there is one
        counts.add(max(1, int(quota)))
    for c in sorted(c for c in counts if 1 <= c <= avail) or [1]:
        session.solve(1, c)  # untimed: the worker pool starts, pages are touched
        frames, t0, phases = 0, time.perf_counter(), [0.0, 0.0, 0.0, 0.0]
        while True:
            ph = session.solve(1, c)
            phases = [a + b for a, b in zip(phases, ph)]
            frames += 1
            el = time.perf_counter() - t0
            if (el >= budget_each and frames >= 2) or frames >= 200:
                break
        curve.append({"threads": c, "value": per_frame * frames / el, "ms_per_frame": 1e3 * el / frames,
                      "phase_ms": {"prepare_integration_responsibilities": 1e3 * phases[0] / frames, "solve": 1e3 * phases[1] / frames,
                                   "integrate_after_substepping": 1e3 * phases[2] / frames},
                      "solve_work_ms_summed_over_workers": 1e3 * phases[3] / frames})
    single = curve[0]["value"]
    for e in curve:
        e["parallel_efficiency"] = e["value"] / (single * e["threads"])
        # where the efficiency goes: the share of threads x Solve time the workers spend inside work blocks (the rest is waiting at the 2 + batches x passes sync
        # stages of every substep), and how much longer the same work blocks take than on one thread (shared memory bandwidth, cache-line traffic on the bodies)
        e["solve_worker_busy_fraction"] = e["solve_work_ms_summed_over_workers"] / (e["threads"] * e["phase_ms"]["solve"])
        e["solve_work_inflation_vs_one_thread"] = e["solve_work_ms_summed_over_workers"] / curve[0]["solve_work_ms_summed_over_workers"]
    best = max(curve, key=lambda e: e["value"])
    # the reported figure: a longer run at the best thread count
    frames, t0 = 0, time.perf_counter()
    while True:
        session.solve(1, best["threads"])
        frames += 1
        el = time.perf_counter() - t0
        if el >= target_seconds / 4.0 or frames >= 400:
            break
    session.close()
    ideal = single * max(1, pin["first_socket_physical_cores"])
    return {"value": max(per_frame * frames / el, best["value"]), "unit": "constraint-iterations/s", "cores": best["threads"], "kind": "port-simd8",
            "single_thread_value": single, "thread_curve": curve, "host_cpus_available": avail,
            "cpu_quota": quota, "cpu_quota_source": quota_path or "no cgroup cpu controller file found",
            "cpu_quota_note": "cgroup CPU quota in CPUs (None = unlimited): threads beyond it share the quota's time slices, which is what a curve that peaks at the quota and "
                              "falls beyond it looks like; `cores` is the thread count of the best MEASURED point, the ideal-socket bound is the figure that does not depend on it",
            "placement": dict(pin, cgroup_cpu_quota=quota, note="workers pinned one per physical core of ONE socket first (sibling threads next, the other socket last); the session's memory "
                                        "first-touched by threads pinned to the same cores (oracle/wide/wide_solver.cpp: PinPlan, AlignedCopy)"),
            "ideal_socket_bound": {"value": ideal, "unit": "constraint-iterations/s",
                                   "note": "the single-thread figure x the physical cores of one socket: what a perfectly scaling socket would reach — the figure to "
                                           "hold the GPU against when the measured curve is distrusted (shared host, sync stages)"},
            "timed_region": "PrepareConstraintIntegrationResponsibilities + Solve + IntegrateAfterSubstepping on a persistent session (aligned library-owned buffers, "
                            "handle sets built at creation): Simulation.cs:278-290, nothing else",
            "sample": f"{ragdolls_sample} ragdolls ({scene.constraint_count} constraints), {frames} frames at {best['threads']} threads after a 1/8/16/32/64/128-thread curve (plus one worker per physical core of the socket), workers pinned, "
                      f"4 substeps x 1 iteration, oracle/wide (C++ AOSOA-8 AVX2 transcription of the reference's CPU path, -O3 -mavx2, no FMA contraction), reference "
                      f"work-block/sync-stage threading, {avail} CPUs available"}


def connected_scene_args(key: str, ragdolls: int):
    return {"pile": ("pile", 100000, 0, 0, 5), "crowd": ("ragdoll_tube", ragdolls, 1, 2, 5)}[key]


def hbm_roofline(kernel: str, launch_us: float, traffic_bytes, algorithmic_bytes: float, stream_bytes=None, **extra):
    """The `roofline` object of the bench line. `frac` is a bandwidth fraction and nothing else: HBM bytes the PMC counters saw per launch (`traffic`) / launch time /
    peak — at most 1 by construction. The SURVEY 8d algorithmic figure (every body gather/scatter of every constraint counted as if it went to HBM) rides beside it
    under algorithmic_*: the island schedules serve body gathers from LDS, so that figure can exceed the peak and is NOT a bandwidth fraction. Without counters
    (N > 1, --no-traffic, no rocprofv3) `achieved` falls back to the compulsory-stream model and says so in `basis`."""
    seconds = launch_us * 1e-6
    basis = "pmc"
    moved = traffic_bytes
    if not moved:
        moved, basis = stream_bytes, "memory_stream_model (no PMC counters in this run)"
    achieved = moved / seconds / 1e9 if moved else None
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
           "traffic": traffic_bytes, "basis": basis, "avg_launch_us": launch_us,
           "algorithmic_bytes_per_launch": algorithmic_bytes, "algorithmic_GBs": algorithmic_bytes / seconds / 1e9,
           "algorithmic_frac_of_peak": algorithmic_bytes / seconds / 1e9 / HBM_PEAK_GBS,
           "algorithmic_note": "SURVEY 8d bytes (per-type figures x constraint-iterations + integration passes); counts body gathers the island schedules serve from LDS, "
                               "so it may exceed the HBM peak — reported for comparison with the survey, not as a bandwidth fraction"}
    if stream_bytes:
        out["memory_stream_bytes_per_launch"] = stream_bytes
        out["memory_stream_GBs"] = stream_bytes / seconds / 1e9
        out["traffic_over_compulsory_stream"] = traffic_bytes / stream_bytes if traffic_bytes else None
    out.update(extra)
    issue = (extra.get("traffic_detail") or {}).get("issue") if isinstance(extra.get("traffic_detail"), dict) else None
    if issue:  # the other ceiling, from the same run's SQ counters: the share of the chip's VALU issue cycles the launch used (`frac` stays the bandwidth fraction)
        out["valu_busy"] = issue["valu_busy"]
    return out


def compulsory_stream_bytes(scene, sd):
    """What a whole-step launch of the island schedule has to move through HBM at least: every body in and out once, the constraint rows once per pass."""
    from bepuphysics2_amd.scene import TYPE_TABLE
    its = sd.iterations()
    stream_bytes = 2 * 128 * scene.body_count
    for batch in scene.batches:
        for tb in batch:
            nb, pf, imf, _name = TYPE_TABLE[tb.type_id]
            refs = (nb + 1) // 2  # the island schedule reads body references as 16-bit halves, two per word
            reads = (refs + pf + imf) * 4 * (sd.substep_count + int(its.sum()))
            writes = imf * 4 * int(its.sum())
            if _name.startswith("Contact"):
                reads += (refs + pf) * 4 * (sd.substep_count - 1)
                writes += int(_name[7]) * 4 * (sd.substep_count - 1)
            stream_bytes += (reads + writes) * tb.count
    return stream_bytes


def connected_scene_leg(name: str, scene_key: str, ragdolls: int, device: int, steps: int = 100, traffic_args=None):
    """A scene that is ONE island (no workgroup's LDS holds it): the general-topology schedule (one launch per batch per stage, hipGraph replay).
    Extra keys on the bench line, never `value`: the pile is BASELINE.json configs[1]; the crowd is configs[2]'s ragdolls lying on each other, as the
    reference's benchmark ends up (RagdollTubeBenchmark.cs:536-569). Roofline: SURVEY.md 8d algorithmic bytes of a step / time of a step."""
    from bepuphysics2_amd.hostlib import HostSimulation
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.roofline import INTEGRATE_BYTES_PER_BODY, FINAL_BYTES_PER_BODY, scene_stage_bytes
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    sim = HostSimulation.scene(*connected_scene_args(scene_key, ragdolls))
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    cb = PoseIntegratorCallbacks()
    solver = HipSolver(device=device, exclusive_device=True)  # this process is alone on the GPU: split plans launch plainly (BEPUHIP_FLAG_EXCLUSIVE_DEVICE) instead of cooperatively
    t0 = time.perf_counter()
    solver.upload(scene)
    upload_ms = 1e3 * (time.perf_counter() - t0)
    unit_state = specialise(solver)
    for _ in range(100):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.reset_state()
    solver.sync()
    quiet_gc()  # (a full collection of Python's garbage collector inside the timed loop would be a third of its time)
    for _ in range(5):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    finite = bool(np.isfinite(solver.get_bodies(scene.body_count)).all())
    solver_family = solver.kernel_family()
    clustered = bool(solver.cluster_cycles().size)
    solver.close()
    its = sd.iterations()
    ws, sv, inc = scene_stage_bytes(scene)
    step_bytes = (sv * int(its.sum()) + ws * sd.substep_count + inc * (sd.substep_count - 1) + INTEGRATE_BYTES_PER_BODY * scene.body_count * sd.substep_count
                  + FINAL_BYTES_PER_BODY * scene.body_count)
    per_step = scene.constraint_count * int((1 + its).sum())
    launches = sd.substep_count * (2 + len(scene.batches) * 1) + len(scene.batches) * int(its.sum()) + 1
    detail = measure_traffic(traffic_args, scene_key) if (traffic_args is not None and clustered) else None
    traffic = detail.get("bytes_per_launch") if isinstance(detail, dict) else None
    return {"workload": f"{name}: {scene.body_count} bodies, {scene.constraint_count} constraints, {len(scene.batches)} batches, ONE island, "
                        f"{sd.substep_count} substeps x {list(map(int, its))} iterations",
            "ms_per_step": ms, "value": per_step / (ms * 1e-3), "unit": "constraint-iterations/s",
            "schedule": "island-per-workgroup, split-island plan (one plain launch per step: BEPUHIP_FLAG_EXCLUSIVE_DEVICE, the bench owns the device; a cooperative launch costs ~25 us more)" if clustered else f"launch-per-batch, hipGraph replay, {launches} launches per step",
            "roofline": hbm_roofline("cluster_kernel<...,SHARED> (whole step in one launch)" if clustered else "whole step (launch-per-batch)", 1e3 * ms if clustered else 1e3 * ms,
                                     traffic, step_bytes, compulsory_stream_bytes(scene, sd) if clustered else None, traffic_detail=detail, working_set_bytes=working_set_bytes(scene),
                                     launch_time_basis="wall time per step (one launch per step)" if clustered else f"whole step, {launches} launches"),
            "upload_ms": upload_ms, "finite": finite, "kernel_family": solver_family, "unit_state": unit_state}


def working_set_bytes(scene) -> int:
    """What a step touches in HBM: every body record and every row of every type batch (references as the island schedule stores them, prestep data, impulses). To be read
    against the 256 MiB Infinity Cache: FETCH_SIZE / WRITE_SIZE count traffic at the fabric side of L2, MALL hits included (MI355X_MICROARCH.md, HBM section)."""
    from bepuphysics2_amd.scene import TYPE_TABLE
    total = 128 * scene.body_count
    for batch in scene.batches:
        for tb in batch:
            nb, pf, imf, _ = TYPE_TABLE[tb.type_id]
            total += ((nb + 1) // 2 + pf + imf) * 4 * tb.count
    return total


def scale_sweep_leg(args, device: int, base_ragdolls: int, factors=(1, 2, 4, 8), steps: int = 50):
    """the headline scene fits the Infinity Cache (136 MB of 256 MiB) and is one workgroup per CU. The same benchmark at 1x / 2x / 4x / 8x the ragdolls
    (1 - 8 M constraints, up to ~1.1 GB of working set, up to eight clusters per CU run back to back): ms/step, constraint-iterations/s, clusters, clusters per CU, the
    working set, the counters' bandwidth fraction (own PMC child runs per size) and the compulsory-stream fraction. Extra keys on the bench line, never `value`."""
    import copy
    import torch
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    cus = int(torch.cuda.get_device_properties(device).multi_processor_count)
    cb = PoseIntegratorCallbacks()
    out = []
    for f in factors:
        ragdolls = base_ragdolls * f
        scene, sd = build_scene(ragdolls, 5)
        solver = HipSolver(device=device)
        t0 = time.perf_counter()
        solver.upload(scene)
        upload_ms = 1e3 * (time.perf_counter() - t0)
        specialise(solver)
        for _ in range(300):  # the headline's pre-warm (clocks, launch policy): the sweep's 1x point is the headline's scene and has to agree with it
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.reset_state()
        solver.sync()
        quiet_gc()
        for _ in range(5):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.sync()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        clusters = int(solver.cluster_cycles().size)
        finite = bool(np.isfinite(solver.get_bodies(scene.body_count)).all())
        solver.close()
        per_step = scene.constraint_count * int((1 + sd.iterations()).sum())
        stream = compulsory_stream_bytes(scene, sd)
        entry = {"ragdolls": ragdolls, "constraints": scene.constraint_count, "bodies": scene.body_count, "ms_per_step": ms, "value": per_step / (ms * 1e-3),
                 "unit": "constraint-iterations/s", "clusters": clusters, "clusters_per_cu": clusters / cus, "working_set_bytes": working_set_bytes(scene),
                 "memory_stream_bytes_per_launch": stream, "memory_stream_frac_of_peak": stream / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "upload_ms": upload_ms, "finite": finite}
        if not args.no_traffic and clusters:
            child = copy.copy(args)
            child.ragdolls = ragdolls
            detail = measure_traffic(child)
            if isinstance(detail, dict) and detail.get("bytes_per_launch"):
                entry["traffic"] = detail["bytes_per_launch"]
                entry["frac"] = detail["bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS  # against the wall time of a step: one launch per step
                entry["traffic_over_compulsory_stream"] = detail["bytes_per_launch"] / stream
            else:
                entry["traffic"] = None
                entry["traffic_detail"] = detail
        out.append(entry)
        del scene
    return {"sizes": out, "cus": cus, "note": "same scene recipe and solve description as the headline at every size; `frac` = PMC traffic (FETCH_SIZE x 2 + WRITE_SIZE, own child "
                                              "runs per size) / wall time of a step / 8 TB/s; a working set above 256 MiB cannot live in the Infinity Cache"}


def optional_leg(fn, *a, **k):
    """An extra leg must never cost the headline its JSON line: a failure is reported in the leg's place."""
    try:
        return fn(*a, **k)
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


EXTRA_LEGS = ("pile", "crowd", "sweep", "widened", "boundary", "lattice")


def run_extra_leg(name: str, args, device: int):
    """One extra leg by name, in this process (the child side of `leg_in_child`)."""
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    targs = None if args.no_traffic else args
    if name == "pile":
        return connected_scene_leg("100k-box pile (BASELINE.json configs[1])", "pile", args.ragdolls, device, traffic_args=targs)
    if name == "crowd":
        return connected_scene_leg(f"{args.ragdolls} ragdolls in contact with their neighbours (configs[2]'s ragdolls, one island)", "crowd", args.ragdolls, device, traffic_args=targs)
    if name == "sweep":
        return scale_sweep_leg(args, device, args.ragdolls)
    if name == "widened":
        return widened_types_leg(args.ragdolls, device)
    if name == "boundary":
        scene, sd = build_scene(args.ragdolls, 5)
        return boundary_leg(scene, sd, PoseIntegratorCallbacks(), device)
    if name == "lattice":
        return lattice_leg(device)
    raise ValueError(name)


def leg_in_child(name: str, args, device: int, timeout: float = 900.0):
    """An extra leg in a process of its own: `optional_leg` catches exceptions, but a leg that takes the PROCESS down (a device fault aborts it) would take the
    headline's JSON line with it — seen once in round 5 (`profiles/r05_s30_bench_fault.txt`: a memory access fault inside the lattice leg, after 205 GPU tests had
    passed on the same library; eight repeats of that leg alone ran clean). The parent has closed its context by then: the child is alone on the GPU, as the legs were
    when they ran in-process. What comes back is the leg's report, or the way the child ended."""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(prefix=f"bepu_leg_{name}_", suffix=".json", dir="/tmp")
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), "--leg-child", name, "--leg-out", path, "--ragdolls", str(args.ragdolls)] + (["--no-traffic"] if args.no_traffic else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BEPU_BENCH_FORCE_DIST")}
    env["BEPU_BENCH_LEG_DEVICE"] = str(device)
    try:
        done = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
        try:
            with open(path) as f:
                return json.load(f)
        except (OSError, ValueError):
            tail = [ln for ln in done.stderr.decode(errors="replace").splitlines() if ln.strip() and "RCCL" not in ln and "ibrccl" not in ln][-2:]
            return {"error": f"leg '{name}' ended with exit code {done.returncode} and no report: " + " | ".join(tail)[:300]}
    except subprocess.TimeoutExpired:
        return {"error": f"leg '{name}' did not finish within {timeout:.0f} s"}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def widened_types_leg(ragdolls: int, device: int, steps: int = 100):
    """The widened constraint types on the driver line (never `value`): the bench scene's ragdolls — same bodies, same constraint graph, same batches — with the seven
    joint types other than BallSocket replaced by widened ones (synthetic.RIG_REMAP: AngularSwivelHinge, DistanceLimit, AngularServo, TwistMotor, AngularAxisMotor, Weld,
    BallSocketServo; seeded random settings). Such a scene runs the second `cluster_kernel` variant (all 44 type ids compiled in)."""
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.scene import TYPE_TABLE, PoseIntegratorCallbacks
    from bepuphysics2_amd.synthetic import RIG_REMAP, rig_scene
    scene, sd = rig_scene(ragdolls)
    cb = PoseIntegratorCallbacks()
    solver = HipSolver(device=device, exclusive_device=True)
    solver.upload(scene)
    specialise(solver)
    for _ in range(60):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.reset_state()
    solver.sync()
    quiet_gc()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.solve(1 / 60, sd, cb, asynchronous=True)
    solver.sync()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    finite = bool(np.isfinite(solver.get_bodies(scene.body_count)).all())
    clusters = int(solver.cluster_cycles().size)
    solver.close()
    its = sd.iterations()
    per_step = scene.constraint_count * int((1 + its).sum())
    out = {"workload": f"{ragdolls} ragdoll rigs: the headline's graph ({scene.constraint_count} constraints, {len(scene.batches)} batches) with "
                       + ", ".join(f"{TYPE_TABLE[a][3]} -> {TYPE_TABLE[b][3]}" for a, b in RIG_REMAP.items()),
           "ms_per_step": ms, "value": per_step / (ms * 1e-3), "unit": "constraint-iterations/s",
           "schedule": f"island-per-workgroup, {clusters} clusters, widened kernel variant" if clusters else "launch-per-batch", "finite": finite,
           # (static, measured on the CPU: profiles/r06_fast_reciprocal_gap.txt, tests/test_fast_reciprocal.py)
           "x86_fast_reciprocal_gap": "CenterDistanceConstraint / CenterDistanceLimit 0.6-1.3e-4, AreaConstraint / VolumeConstraint ~1e-5 relative velocity error after 8 substeps against "
                                      "vrcpps / vrsqrtps (the reference's MathHelper.FastReciprocal on an AVX host); every other type bit-identical to both oracles"}
    # The worst case of the work-item design, every round: all 44 type ids drawn at random in 4,000 islands of 16 bodies with 64 constraints — 790 type
    # batches, one or two constraints per cluster and type batch, i.e. ~1.3 of 64 lanes busy per work item and ~6,300 work items per cluster and step.
    try:
        tests_on_path()
        import small_scenes  # (test infrastructure: the generator of tools/perf_widened.py)
        from bepuphysics2_amd.scene import SolveDescription
        everything = small_scenes.island_scene(11, 4000, 16, 64, sorted(TYPE_TABLE))
        sd44 = SolveDescription(1, 4)
        solver = HipSolver(device=device, exclusive_device=True)
        solver.upload(everything)
        for _ in range(40):
            solver.solve(1 / 60, sd44, cb, asynchronous=True)
        solver.reset_state()
        solver.sync()
        quiet_gc()
        t0 = time.perf_counter()
        for _ in range(50):
            solver.solve(1 / 60, sd44, cb, asynchronous=True)
        solver.sync()
        ms44 = 1e3 * (time.perf_counter() - t0) / 50
        out["all_types_ms"] = ms44
        out["all_types_value"] = everything.constraint_count * int((1 + sd44.iterations()).sum()) / (ms44 * 1e-3)
        out["all_types_workload"] = f"all 44 type ids at random: {everything.body_count} bodies, {everything.constraint_count} constraints, {len(everything.batches)} batches, {int(solver.cluster_cycles().size)} clusters"
        out["all_types_finite"] = bool(np.isfinite(solver.get_bodies(everything.body_count)).all())
        solver.close()
    except Exception as e:  # noqa: BLE001
        out["all_types_error"] = f"{type(e).__name__}: {e}"[:200]
    return out


def boundary_leg(scene, sd, cb, device: int):
    """What a C# host pays around the resident-in-HBM rate (never `value`). All host buffers are registered once (BufferPool blocks are pinned memory that lives as
    long as the simulation, BufferPool.cs:42,83) and the legs are:
    * a full upload: set_bodies + begin/set/end (references converted and clusters planned on the host; prestep data and impulses copied as they are and transposed
      on the device);
    * the RESIDENT frame a HipTimestepper runs once the scene is on the device: every contact type batch's prestep data refreshed (what the narrow phase rewrites
      every frame, NarrowPhaseConstraintUpdate.cs:147-207), solve, poses + velocities back into the host's BodyDynamics array — all asynchronous, one sync;
    * the full round trip of round 2 for comparison (every body in, every body, impulse and prestep row back);
    * structural churn through the add / remove calls."""
    from bepuphysics2_amd.native import HipSolver, _check, _ptr
    from bepuphysics2_amd.scene import TYPE_TABLE
    solver = HipSolver(device=device)
    work = scene.copy()
    solver.register_host_memory(work.bodies)
    for b in work.batches:
        for tb in b:
            if tb.count:
                solver.register_host_memory(tb.prestep)
                solver.register_host_memory(tb.accumulated)
    solver.upload(work)  # untimed: the first upload also allocates the staging buffers a simulation keeps
    kin = np.ascontiguousarray(scene.constrained_kinematic_indices(), dtype=np.int32)
    body_ms, upload_ms = [], []
    for _ in range(3):  # the median of three re-uploads (one sample moved by 3 ms from run to run on the pool's hosts)
        t0 = time.perf_counter()
        solver.set_bodies(work.bodies)
        t1 = time.perf_counter()
        solver.set_constraints(work, sd.fallback_batch_threshold)
        _check(solver.lib, solver.lib.bepuhip_set_constrained_kinematics(solver.ctx, _ptr(kin), kin.size))
        t2 = time.perf_counter()
        body_ms.append(1e3 * (t1 - t0)); upload_ms.append(1e3 * (t2 - t1))
    solver._scene_meta = [(bi, tb.type_id, tb.count) for bi, b in enumerate(work.batches) for tb in b]
    out = {"set_bodies_ms": sorted(body_ms)[1], "end_constraints_ms": sorted(upload_ms)[1],
           "end_constraints_note": "begin / set_type_batch x type batches / end of an upload that re-uses the context's staging buffers: body references AOSOA -> rows and island "
                                   "(cluster) planning on the host; prestep data and impulses copied as they are (registered memory) and transposed on the device"}
    frames = 20
    from bepuphysics2_amd import native as nat
    contact_tbs = [(bi, tb) for bi, b in enumerate(work.batches) for tb in b if tb.count and TYPE_TABLE[tb.type_id][3].startswith("Contact")]
    joint_tbs = [(bi, tb) for bi, b in enumerate(work.batches) for tb in b if tb.count and not TYPE_TABLE[tb.type_id][3].startswith("Contact")]
    # the frame's tables, prepared once (a host keeps them between frames as long as the type batches' buffers stay where they are)
    prestep_in = solver.row_transfer_table([(nat.ROWS_UPDATE_PRESTEP, bi, tb.type_id, 0, tb.prestep.reshape(-1)) for bi, tb in contact_tbs])
    keep = [solver._transfer_keepalive]
    shim_in = solver.row_transfer_table([(kind, bi, tb.type_id, 0, (tb.prestep if kind == nat.ROWS_UPDATE_PRESTEP else tb.accumulated).reshape(-1))
                                         for bi, tb in contact_tbs for kind in (nat.ROWS_UPDATE_PRESTEP, nat.ROWS_UPDATE_IMPULSES)])
    keep.append(solver._transfer_keepalive)
    shim_out = solver.row_transfer_table([(nat.ROWS_GET_IMPULSES, bi, tb.type_id, 0, tb.accumulated.reshape(-1)) for bi, tb in contact_tbs + joint_tbs])
    keep.append(solver._transfer_keepalive)
    solver.solve(1 / 60, sd, cb)

    def timed(fn, n=frames):
        fn()
        solver.sync()
        quiet_gc()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
            solver.sync()
        return 1e3 * (time.perf_counter() - t0) / n

    def frame():
        solver.transfer_rows(prestep_in)
        solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.get_poses_and_velocities(work.bodies, asynchronous=True)

    def shim_frame():
        solver.set_bodies(work.bodies)
        solver.transfer_rows(shim_in)
        solver.solve(1 / 60, sd, cb, asynchronous=True)
        solver.get_poses_and_velocities(work.bodies, asynchronous=True)
        solver.transfer_rows(shim_out)

    out["frame_through_abi_ms"] = timed(frame)
    out["frame_stages_ms"] = {"contact_prestep_in": timed(lambda: solver.transfer_rows(prestep_in)),
                              "solve": timed(lambda: solver.solve(1 / 60, sd, cb, asynchronous=True)),
                              "poses_and_velocities_out": timed(lambda: solver.get_poses_and_velocities(work.bodies, asynchronous=True)),
                              "empty_sync": timed(lambda: None)}
    out["frame_through_abi_note"] = (f"the resident frame: ONE bepuhip_transfer_rows_async for all {len(contact_tbs)} contact type batches' prestep data "
                                     f"({sum(tb.prestep.nbytes for _, tb in contact_tbs) / 1e6:.1f} MB, read from the registered host buffers by the transposing kernel itself) + solve_async + "
                                     f"get_poses_and_velocities_async ({work.bodies.shape[0] * 64 / 1e6:.1f} MB, written into the host's BodyDynamics array by a kernel) + one sync; "
                                     "frame_stages_ms: each stage on its own, with its own sync (round 4, one copy + two launches per type batch and a strided 2-D copy: 2.8-2.9 ms)")
    out["shim_frame_ms"] = timed(shim_frame)
    out["shim_frame_stages_ms"] = {"set_bodies": timed(lambda: solver.set_bodies(work.bodies)), "contact_prestep_and_impulses_in": timed(lambda: solver.transfer_rows(shim_in)),
                                   "all_impulses_out": timed(lambda: solver.transfer_rows(shim_out))}
    out["shim_frame_note"] = (f"what integration/csharp/HipTimestepper.cs moves by default every frame: every body in ({work.bodies.nbytes / 1e6:.1f} MB), the contacts' prestep data and "
                              f"impulses in, solve, poses and velocities out, the accumulated impulses of ALL {len(contact_tbs) + len(joint_tbs)} type batches out "
                              f"({sum(tb.accumulated.nbytes for _, tb in contact_tbs + joint_tbs) / 1e6:.1f} MB); the host-side diff and comparison are not in it")
    t0 = time.perf_counter()
    for _ in range(3):
        solver.set_bodies(work.bodies)
        solver.solve(1 / 60, sd, cb)
        solver.download(work)
    out["full_round_trip_frame_ms"] = 1e3 * (time.perf_counter() - t0) / 3
    out["full_round_trip_note"] = "round 2's frame_through_abi_ms: set_bodies + solve + get_bodies + get_accumulated_impulses / get_prestep of every type batch, synchronous calls"
    # structural churn: the last 1 % of every two-body contact type batch removed and added again (same bodies, same prestep) per frame — what the narrow phase does to
    # pairs whose manifold changed. The constraint set is the same after every frame, so the frames are comparable.
    contact = [(bi, tb) for bi, b in enumerate(scene.batches) for tb in b if TYPE_TABLE[tb.type_id][3].startswith("Contact") and tb.bodies == 2 and tb.count > 100]
    if contact:
        picks = []
        for bi, tb in contact:
            k = max(1, tb.count // 100)
            refs, pre = tb.refs_lanes(scene.bundle_width), tb.prestep_lanes(scene.bundle_width)
            picks.append((bi, tb.type_id, tb.count, [(refs[i].copy(), pre[i].copy()) for i in range(tb.count - k, tb.count)]))
        calls = sum(len(p[3]) for p in picks)

        def churn():
            for bi, t, count, lanes in picks:
                for j in range(len(lanes)):
                    solver.remove_constraint(bi, t, count - 1 - j)
                for refs, pre in lanes:
                    solver.add_constraint(bi, t, refs, pre)

        # the same frame as ONE bepuhip_apply_structural_ops call (round 4): the operation table is what a host builds from its own structures (the C# shim's diff)
        rows, words, at = [], [], 0
        for bi, t, count, lanes in picks:
            for j in range(len(lanes)):
                rows.append((1, bi, t, count - 1 - j, 0, 0, 0, 0))
            for j, (refs, pre) in enumerate(lanes):
                rows.append((0, bi, t, count - len(lanes) + j, 0, 0, at, 0))
                words += [np.ascontiguousarray(refs, dtype=np.int32).view(np.uint32), np.ascontiguousarray(pre, dtype=np.float32).view(np.uint32)]
                at += refs.size + pre.size
        table, payload = np.asarray(rows, dtype=np.int32), np.concatenate(words)

        churn()
        solver.solve(1 / 60, sd, cb)
        t0 = time.perf_counter()
        for _ in range(frames):
            churn()
            solver.solve(1 / 60, sd, cb)
        out["structural_frame_per_call_ms"] = 1e3 * (time.perf_counter() - t0) / frames
        t_calls = 0.0
        t0 = time.perf_counter()
        for _ in range(frames):
            t1 = time.perf_counter()
            solver.apply_structural_op_table(table, payload)
            t_calls += time.perf_counter() - t1
            solver.solve(1 / 60, sd, cb)
        out["structural_frame_ms"] = 1e3 * (time.perf_counter() - t0) / frames
        out["structural_ops_call_ms"] = 1e3 * t_calls / frames
        t0 = time.perf_counter()
        for _ in range(frames):
            solver.solve(1 / 60, sd, cb)
        out["solve_after_structural_updates_ms"] = 1e3 * (time.perf_counter() - t0) / frames
        stayed = bool(solver.cluster_cycles().size)
        out["structural_frame_note"] = (f"{calls} removals + {calls} additions (1 % of the two-body contacts) in ONE bepuhip_apply_structural_ops call + solve "
                                        f"(structural_ops_call_ms: the call alone; structural_frame_per_call_ms: the same frame as {2 * calls} single calls from Python, round 3's form); the context "
                                        + ("stayed on the island schedule (freed device slots reused, the predecessor lists of the touched clusters rebuilt on the host)" if stayed else
                                           "left the island schedule for the launch-per-batch one"))
        # a re-plan beside the frames (round 6, bepuhip_replan_begin / _commit) against bepuhip_replan: what each costs the thread that runs the frames
        t0 = time.perf_counter()
        solver.replan()
        out["replan_sync_ms"] = 1e3 * (time.perf_counter() - t0)
        solver.solve(1 / 60, sd, cb)
        t0 = time.perf_counter()
        solver.replan_begin()
        out["replan_begin_ms"] = 1e3 * (time.perf_counter() - t0)
        in_flight = 0
        while True:
            t0 = time.perf_counter()
            committed = solver.replan_commit(wait=in_flight >= 200)
            commit_ms = 1e3 * (time.perf_counter() - t0)
            if committed:
                break
            solver.apply_structural_op_table(table, payload)
            solver.solve(1 / 60, sd, cb)
            in_flight += 1
        out["replan_commit_ms"] = commit_ms
        out["replan_blocking_ms"] = out["replan_begin_ms"] + commit_ms
        out["replan_frames_in_flight"] = in_flight
        out["replan_schedule_after"] = solver.schedule()
        out["replan_note"] = (f"bepuhip_replan blocks the caller for the host planner (replan_sync_ms); begin + commit block it for the snapshot, the tables' upload and the replay of the "
                              f"{in_flight} x {2 * calls} structural operations of the frames solved while the worker planned (replan_blocking_ms)")
    solver.close()
    return out


def lattice_leg(device: int, ragdolls: int = 2000, world: int = 2, frames: int = 2):
    """configs[4]'s split at N=1: one connected lattice cut into `world` shares that all run on this GPU (one thread and one context per rank, exchange through host
    memory), compared with the unsplit solve of the same library. Reports what each exchange mode costs in accuracy; the timings of this leg mean nothing."""
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.hostlib import HostSimulation
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    try:
        sim = HostSimulation.scene("ragdoll_tube", ragdolls, 1, 1, 5)
        scene, sd = sim.export(), sim.solve_description()
        sim.close()
        cb = PoseIntegratorCallbacks()
        ref = scene.copy()
        solver = HipSolver(device=device)
        solver.upload(ref, sd.fallback_batch_threshold)
        for _ in range(frames):
            solver.solve(1 / 60, sd, cb)
        solver.download(ref)
        solver.close()
        vel = [8, 9, 10, 12, 13, 14]
        scale = float(np.abs(ref.bodies[:, vel]).max())
        owner = lattice.owner_by_groups(scene, world, 16)
        out = {"workload": f"one connected lattice of {ragdolls} ragdolls ({scene.constraint_count} constraints, {len(scene.batches)} batches) cut into {world} shares on this GPU, "
                           f"{frames} frames, error = max over bodies of |v_split - v_unsplit| / max|v_unsplit| against the unsplit solve of the same library"}
        for name, exact in (("per_pass_block_jacobi", False), ("per_batch_exact", True)):
            shares = [lattice.make_share(scene, owner, r, world, mass_split=not exact) for r in range(world)]
            # (the per-pass mode runs each share on its island plan, one launch per pass; the per-batch mode needs an exchange after every batch: launch-per-batch)
            ex = lattice.solve_shares_in_process(lambda: HipSolver(device=device, use_clusters=not exact), shares, 1 / 60, sd, cb, frames=frames, exact=exact)
            merged = lattice.merge_owned(scene, shares)
            per_body = np.abs(ref.bodies[:, vel] - merged.bodies[:, vel]).max(axis=1) / scale
            out[name] = {"within_north_star_tolerance": bool(float(per_body.max()) <= 1e-4),  # BASELINE.json: <= 1e-4 relative velocity error
                         "velocity_err_max": float(per_body.max()), "velocity_err_median": float(np.median(per_body)),
                         "bit_identical": bool(np.array_equal(ref.bodies[:, :15].view(np.int32), merged.bodies[:, :15].view(np.int32))),
                         "exchanges_per_frame": ex.calls // frames, "boundary_bodies": int(shares[0].boundary_total)}
        # (all members of the group share THIS device here, and clusters that wait for each other must all be resident: the plan is held to 120 clusters in total —
        # on a multi-GPU node every device gets a full round of its own, bench.py --lattice --lattice-exact)
        saved = os.environ.get("BEPUHIP_SPLIT_CLUSTERS")
        os.environ["BEPUHIP_SPLIT_CLUSTERS"] = "120"
        try:
            grouped = lattice.solve_group_in_process(lambda: HipSolver(device=device, exclusive_device=True), scene, world, 1 / 60, sd, cb, frames=frames)
        finally:
            if saved is None:
                os.environ.pop("BEPUHIP_SPLIT_CLUSTERS", None)
            else:
                os.environ["BEPUHIP_SPLIT_CLUSTERS"] = saved
        per_body = np.abs(ref.bodies[:, vel] - grouped.bodies[:, vel]).max(axis=1) / scale
        out["device_group_exact"] = {"within_north_star_tolerance": bool(float(per_body.max()) <= 1e-4), "velocity_err_max": float(per_body.max()),
                                     "bit_identical": bool(np.array_equal(ref.bodies[:, :15].view(np.int32), grouped.bodies[:, :15].view(np.int32))),
                                     "exchanges_per_frame": 1, "schedules": [info[0] for info in grouped.group_info]}
        out["configs4_mode"] = ("device_group_exact (round 5): one plan, every device runs its clusters in ONE island-kernel launch per step, shared bodies through records pushed into "
                                "every device's table, one all-reduce per frame — bit-identical to the unsplit solve, like per_batch_exact (136 collectives per frame on the "
                                "launch-per-batch schedule); the per-pass block-Jacobi mode is an approximation outside BASELINE.json's 1e-4 tolerance, NOT a configs[4] result")
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:300]}


def measure_traffic(args, scene_key: str = "main"):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected as MI355X_MICROARCH.md's HBM section prescribes: separate
    rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), counters in KiB, and the gfx950 correction (FETCH_SIZE
    reports half of a wide coalesced read stream: doubled here; WRITE_SIZE is uncalibrated and reported as is). Returns None when rocprofv3
    is unavailable or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    out = {}
    try:
        dominant = None
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE")):
            d = tempfile.mkdtemp(prefix="bepu_pmc_", dir="/tmp")
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BEPU_BENCH_FORCE_DIST")}
            env["TMPDIR"] = "/tmp"
            sq_pass = counters[0] == "SQ_ACTIVE_INST_VALU"  # cycle counters want the launches of a settled launch policy and warm caches: the last five of thirty
            cmd = ["rocprofv3", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "5" if sq_pass else "3", "--warmup", "25" if sq_pass else "1", "--ragdolls", str(args.ragdolls), "--no-cpu-baseline", "--no-traffic", "--no-prewarm",
                   "--traffic-child", scene_key]
            done = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=not sq_pass)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            per_kernel = {}
            rows = [r for f in files for r in csv.DictReader(open(f)) if r["Counter_Name"] in counters]
            if sq_pass and dominant:  # the last five step launches only
                ids = sorted({int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"] == dominant})[-5:]
                rows = [r for r in rows if r["Kernel_Name"] == dominant and int(r["Dispatch_Id"]) in ids]
            for r in rows:
                a = per_kernel.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
            shutil.rmtree(d, ignore_errors=True)
            if sq_pass:  # what the instruction issue looks like next to the bytes (optional: a failed pass leaves the traffic figures standing)
                sq = {c: per_kernel[(dominant, c)][0] / per_kernel[(dominant, c)][1] for c in counters if (dominant, c) in per_kernel}
                if done.returncode == 0 and len(sq) == len(counters):
                    out["SQ"] = sq
                continue
            dom = max(per_kernel.items(), key=lambda kv: kv[1][0]) if per_kernel else None
            if dom is None:
                return None
            dominant = dominant or dom[0][0]
            out[counters[0]] = {"kernel": dom[0][0][:80], "KiB_per_launch": dom[1][0] / dom[1][1]}
        fetch = out["FETCH_SIZE"]["KiB_per_launch"] * 1024.0
        write = out["WRITE_SIZE"]["KiB_per_launch"] * 1024.0
        result = {"bytes_per_launch": 2.0 * fetch + write, "fetch_bytes_raw": fetch, "fetch_bytes_gfx950_corrected": 2.0 * fetch, "write_bytes": write,
                  "kernel": out["FETCH_SIZE"]["kernel"], "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950); WRITE_SIZE as reported "
                  "(calibrated on this kernel's stores, profiles/r04_s12_write_size_probe.txt: exact for coalesced rows and for records written as lane pairs, 64 B per lone 16-byte store; r04_s18_access_size_probe.txt: a record poll is one 64-byte request, "
                  "which the x2 books at 128 — on split plans `traffic` therefore overstates what crosses the fabric by the polls' raw bytes)"}
        if "SQ" in out:
            # All four are sums over the chip: SQ_* in quad-cycles over all waves, GRBM_GUI_ACTIVE in cycles over the 8 XCDs (value / 8 / launch time = the shader clock).
            import torch
            xcds, simds = 8, 4 * torch.cuda.get_device_properties(0).multi_processor_count
            sq = out["SQ"]
            simd_quad_cycles = sq["GRBM_GUI_ACTIVE"] / xcds / 4.0 * simds
            result["issue"] = {"valu_busy": sq["SQ_ACTIVE_INST_VALU"] / simd_quad_cycles, "waves_per_simd": sq["SQ_WAVE_CYCLES"] / simd_quad_cycles,
                               "wave_time_waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], "cycles_per_launch": sq["GRBM_GUI_ACTIVE"] / xcds,
                               "method": "rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE (own pass); valu_busy = VALU-issuing quad-cycles of all "
                                         "waves / quad-cycles the chip's SIMDs offer during the launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs)"}
        return result
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}


def traffic_child(args, device: int):
    """What the rocprofv3 --pmc passes of measure_traffic run: a few solves of one scene, nothing else (no torch, no timing)."""
    from bepuphysics2_amd.hostlib import HostSimulation
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    if args.traffic_child == "main":
        scene, sd = build_scene(args.ragdolls, 5)
    else:
        sim = HostSimulation.scene(*connected_scene_args(args.traffic_child, args.ragdolls))
        scene, sd = sim.export(), sim.solve_description()
        sim.close()
    solver = HipSolver(device=device, exclusive_device=True)
    solver.upload(scene)
    cb = PoseIntegratorCallbacks()
    for _ in range(args.warmup):  # synchronous: the launch policy settles on completed samples (a queue of thirty asynchronous solves would run out before the first one ends)
        solver.solve(1.0 / 60.0, sd, cb)
    for _ in range(args.steps):
        solver.solve(1.0 / 60.0, sd, cb, asynchronous=True)
    solver.sync()
    solver.close()


def run_lattice_group(args, rank, local_rank, world, dist, torch, scene, sd, standalone=True):
    """configs[4], exact, on the island schedule (round 5): a device group (bepuhip_set_device_group). Every rank uploads the whole lattice, plans the same clusters and runs
    its range of them in ONE launch per step; shared bodies cross ranks through the split plan's records, pushed into every rank's table over the fabric; one all-reduce of
    the owned bodies per step. Strong scaling: the scene is fixed."""
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.roofline import INTEGRATE_BYTES_PER_BODY, FINAL_BYTES_PER_BODY, scene_stage_bytes
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    cb = PoseIntegratorCallbacks()
    solver = HipSolver(device=local_rank, exclusive_device=True)
    solver.set_device_group(world, rank)
    solver.upload(scene, sd.fallback_batch_threshold)
    schedule = {0: "launch-per-batch schedule", 1: "island schedule (whole islands), one launch per step", 2: "island schedule on a split-island plan, one launch per step"}[solver.schedule()]
    line = None
    if dist is not None and world > 1:
        handles = [None] * world
        dist.all_gather_object(handles, solver.export_shared_records())  # hipIpcMemHandle_t of every rank's record table
        for k, r in enumerate(r for r in range(world) if r != rank):
            solver.import_peer_records(k, handles[r])
        ids = [solver.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        solver.comm_init(ids[0], rank, world)
        dist.barrier()  # every table is cleared and mapped before the first record is pushed
    dt = 1.0 / 60.0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        solver.sync()

    def step():
        solver.solve(dt, sd, cb, asynchronous=True)
        solver.sync_owned_bodies()

    # (round 6: 30 steps for the clocks and the launch policy's fifteen samples — the first ten steps of a context run 8-15 % slow; the 300 of round 5 were there for a
    # "cold-start anomaly" that was Python's garbage collector: quiet_gc)
    prewarm_steps = 0 if args.no_prewarm else 30
    quiet_gc()
    for _ in range(prewarm_steps):
        step()
    if prewarm_steps:
        solver.reset_state()
        solver.sync()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        its = sd.iterations()
        units = scene.constraint_count * int((1 + its).sum()) * args.steps
        ws_bytes, sv_bytes, inc_bytes = scene_stage_bytes(scene)
        step_bytes = (sv_bytes * int(its.sum()) + ws_bytes * sd.substep_count + inc_bytes * (sd.substep_count - 1)
                      + INTEGRATE_BYTES_PER_BODY * scene.body_count * sd.substep_count + FINAL_BYTES_PER_BODY * scene.body_count)
        achieved = step_bytes / (elapsed / args.steps) / 1e9
        line = ({
            "metric": "constraint-iterations/sec", "value": units / elapsed, "unit": "constraint-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "one connected ragdoll lattice (BASELINE.json configs[4]): "
                                   f"{args.ragdolls} ragdolls, {scene.constraint_count} constraints, {scene.body_count} bodies, {len(scene.batches)} batches, "
                                   f"{sd.substep_count} substeps x {sd.velocity_iteration_count} velocity iteration(s), dt=1/60, {world} device(s)",
                       "sharding": f"device group of {world}: one plan, each device runs {int(solver.cluster_cycles().size) // max(world, 1)} of its {int(solver.cluster_cycles().size)} clusters; shared bodies through "
                                   "event-numbered records pushed into every device's table (exact: bit-identical to one device); "
                                   + ("one ncclAllReduce of the owned bodies' MotionState per step" if world > 1 else "no collective on one device") + f"; {schedule}",
                       "exchanges_per_step": 1 if world > 1 else 0, "finite": bool(np.isfinite(solver.get_bodies(scene.body_count)).all()),
                       "device_prewarm": f"{prewarm_steps} untimed steps during setup, uploaded state restored before the {args.warmup} warm-up steps"},
            "roofline": {"bound": "hbm", "kernel": f"whole step ({schedule}), algorithmic bytes", "achieved": achieved, "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None},
            "cpu_baseline": None})
        line["config"]["schedule_is_island"] = solver.schedule() == 2  # (False: the plan did not fit — more clusters per device than CUs, or LDS — and the group ran launch-per-batch)
        if standalone:
            emit_line(line)
    if dist is not None:
        dist.barrier()
        if standalone:
            dist.destroy_process_group()
    solver.close()
    return line if rank == 0 else None


def run_lattice(args, rank, local_rank, world, dist, torch):
    """configs[4]: one connected lattice, shares + ghosts + RCCL exchange (bepuphysics2_amd/lattice.py). Strong scaling: the scene is fixed."""
    from bepuphysics2_amd import lattice
    from bepuphysics2_amd.hostlib import HostSimulation
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.roofline import INTEGRATE_BYTES_PER_BODY, FINAL_BYTES_PER_BODY, scene_stage_bytes
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks
    sim = HostSimulation.scene("ragdoll_tube", args.ragdolls, 1, 1, 5)  # every rank builds the same scene and cuts out its share
    scene, sd = sim.export(), sim.solve_description()
    sim.close()
    exact = bool(args.lattice_exact)
    if exact and not args.lattice_per_batch:
        return run_lattice_group(args, rank, local_rank, world, dist, torch, scene, sd)
    share = lattice.make_share(scene, lattice.owner_by_groups(scene, world, 16), rank, world, mass_split=not exact)
    cb = PoseIntegratorCallbacks()
    solver = HipSolver(device=local_rank, use_clusters=not exact and not args.lattice_no_clusters, exclusive_device=True)
    solver.upload(share.scene, sd.fallback_batch_threshold)
    schedule = {0: "launch-per-batch schedule", 1: "island schedule, one launch per pass (whole islands)", 2: "island schedule on a split-island plan, one launch per pass"}[solver.schedule()]
    solver.set_boundary_bodies(share.boundary_local)
    solver.set_boundary_layout(share.boundary_slot, share.boundary_total, None if exact else share.boundary_holders)
    solver.set_exchange_mode(1 if exact else 0)
    if dist is not None:  # the library's own communicator: rank 0 makes the id, torch.distributed only carries its 128 bytes
        ids = [solver.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        solver.comm_init(ids[0], rank, world)
    dt = 1.0 / 60.0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    prewarm_steps = 0 if args.no_prewarm else 30  # (see run_lattice_group)
    quiet_gc()
    for _ in range(prewarm_steps):
        solver.solve_lattice(dt, sd, cb)
    if prewarm_steps:
        solver.reset_state()
        solver.sync()
    for _ in range(args.warmup):
        solver.solve_lattice(dt, sd, cb)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.solve_lattice(dt, sd, cb)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        its = sd.iterations()
        units = scene.constraint_count * int((1 + its).sum()) * args.steps
        ws_bytes, sv_bytes, inc_bytes = scene_stage_bytes(scene)
        step_bytes = (sv_bytes * int(its.sum()) + ws_bytes * sd.substep_count + inc_bytes * (sd.substep_count - 1)
                      + INTEGRATE_BYTES_PER_BODY * scene.body_count * sd.substep_count + FINAL_BYTES_PER_BODY * scene.body_count)
        achieved = step_bytes / (elapsed / args.steps) / 1e9
        emit_line({
            "metric": "constraint-iterations/sec", "value": units / elapsed, "unit": "constraint-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "one connected ragdoll lattice (BASELINE.json configs[4]): "
                                   f"{args.ragdolls} ragdolls, {scene.constraint_count} constraints, {scene.body_count} bodies, {len(scene.batches)} batches, "
                                   f"{sd.substep_count} substeps x {sd.velocity_iteration_count} velocity iteration(s), dt=1/60, split into {world} share(s)",
                       "sharding": f"bodies by ragdoll ranges, {share.boundary_total} boundary bodies, "
                                   + (f"exact per-batch exchange of XOR bit patterns ({int((1 + its).sum()) * len(scene.batches)} " if exact else
                                      f"mass-split block-Jacobi exchange after every pass ({int((1 + its).sum())} ")
                                   + f"ncclAllReduce of {share.boundary_total * 24} bytes per step, enqueued on the solver's stream by bepuhip_solve_lattice: no host "
                                     f"synchronisation inside a frame); {schedule}",
                       "exchanges_per_step": int((1 + its).sum()) * (len(scene.batches) if exact else 1)},
            "roofline": {"bound": "hbm", "kernel": f"whole step ({schedule} + exchanges), algorithmic bytes", "achieved": achieved, "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None},
            "cpu_baseline": None})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    solver.close()


def lattice_group_command(args, world: int):
    """The command line of one rank of the configs[4] leg (a standalone `--lattice --lattice-exact` run: run_lattice_group prints its own line on rank 0). Pure."""
    return [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--lattice", "--lattice-exact", "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--ragdolls", str(args.ragdolls)] + (["--no-prewarm"] if args.no_prewarm else [])


def lattice_group_in_children(args, rank: int, local_rank: int, world: int, dist, timeout: float = 420.0):
    """The device-group leg of an N > 1 run, one child process per rank (same isolation as `leg_in_child`): rank 0 picks a free port, every rank starts
    `bench.py --lattice --lattice-exact` with its own RANK / LOCAL_RANK and that port, waits for it (bounded), and rank 0 reads the child's JSON line. The parents keep
    their process group idle meanwhile and meet at a barrier afterwards. Returns the leg's line on rank 0 (or how the children ended), None elsewhere."""
    import socket
    port = [None]
    if rank == 0:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port[0] = sock.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
              "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "BEPU_BENCH_GROUP_LEG"):
        env.pop(k, None)  # (the children are a job of their own: nothing of the launcher's store or agent)
    if world == 1:
        env["BEPU_BENCH_FORCE_DIST"] = "1"  # (the one-GPU rehearsal of this path: BEPU_BENCH_GROUP_LEG=1)
    report = None
    try:
        done = subprocess.run(lattice_group_command(args, world), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        status = {"rank": rank, "exit_code": done.returncode}
        if rank == 0:
            lines = [ln for ln in done.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            try:
                report = json.loads(lines[-1]) if lines else None
            except ValueError:
                report = None
        if done.returncode != 0 or (rank == 0 and report is None):
            status["stderr_tail"] = " | ".join([ln for ln in done.stderr.decode(errors="replace").splitlines() if ln.strip()][-2:])[:300]
    except subprocess.TimeoutExpired:
        status = {"rank": rank, "error": f"child did not finish within {timeout:.0f} s (killed)"}
    statuses = [None] * world
    dist.all_gather_object(statuses, status)
    if rank != 0:
        return None
    failed = [st for st in statuses if st.get("exit_code", 1) != 0]
    if report is None or failed:
        return {"error": "the device-group leg did not complete on every rank", "ranks": failed or statuses, "partial": report}
    return report


_LINE_FD = None


def claim_stdout():
    """ONE JSON line on stdout and nothing else: RCCL prints a five-line version banner on the C library's stdout when its first communicator comes up (seen after the
    JSON line in a pipe: C stdio flushes at exit), and whatever else a library of the process writes there would land beside the line too. From here on file descriptor 1
    of this process (and of what it starts) is stderr; `emit_line` writes the line to the descriptor stdout used to be."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(obj):
    text = (json.dumps(obj) + "\n").encode()
    if _LINE_FD is None:
        sys.stdout.write(text.decode())
        sys.stdout.flush()
    else:
        os.write(_LINE_FD, text)


def quiet_gc():
    """Before a timed region: one full collection now, and everything alive moved to the permanent generation. CPython's cyclic collector runs a full collection every few
    thousand container allocations; with torch imported that is a 35 ms pause on the host — which, landing inside a timed loop of 50 asynchronous 0.25 ms steps, was round
    5's "cold-start anomaly" (the first sixty lattice steps at a third of the settled rate: profiles/r06_s6_cold_start_is_python_gc.txt). The device and the library never
    see it; a C# host has no such collector on this path (the solve is one P/Invoke)."""
    import gc
    gc.collect()
    gc.freeze()


def launch_plan(gpus: int, env: dict, device_count: int, argv: list):
    """What `python bench.py --gpus N` does about its ranks (it used to run ONE rank and print n_gpus 1 when N > 1 and no launcher had set WORLD_SIZE).
    ("run", None): this process is a rank (N = 1, or started by torchrun with WORLD_SIZE = N) | ("spawn", command): no launcher — start the N ranks ourselves, the way the
    driver does | ("error", text): a world that does not match, or more ranks than devices. Pure: tested on the CPU (tests/test_bench_launcher.py)."""
    world = int(env.get("WORLD_SIZE", "1"))
    if gpus < 1:
        return "error", f"--gpus {gpus}: at least one GPU"
    if "WORLD_SIZE" in env and world != gpus:
        return "error", f"--gpus {gpus} != WORLD_SIZE {world}: the launcher and the flag must agree"
    if gpus == 1 or "WORLD_SIZE" in env:
        return "run", None
    if device_count < gpus:
        return "error", f"--gpus {gpus} but this node shows {device_count} GPU(s): one rank per GPU (RCCL admits one rank per device)"
    port = env.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    return "spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", port,
                     os.path.abspath(__file__)] + list(argv)


def multi_gpu_self_checks(torch, dist, world: int, local_rank: int) -> dict:
    """Start-up checks of an N > 1 run, with errors that say what is wrong instead of a hang inside the first collective: every pair of devices can reach each other
    (the device group's records are pushed into the peers' memory), the communicator has as many ranks as the job, one device per rank."""
    report = {"world": world, "devices_visible": int(torch.cuda.device_count())}
    if dist.get_world_size() != world:
        raise SystemExit(f"process group of {dist.get_world_size()} ranks for --gpus {world}")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} visible device(s): RCCL admits one rank per device")
    unreachable = [(a, b) for a in range(world) for b in range(world) if a != b and not torch.cuda.can_device_access_peer(a, b)]
    report["peer_access_all_pairs"] = not unreachable
    if unreachable:
        raise SystemExit(f"devices without peer access: {unreachable[:8]} - the device group pushes records into its peers' memory over xGMI")
    probe = torch.ones(1, device=f"cuda:{local_rank}")
    dist.all_reduce(probe)
    if int(probe.item()) != world:
        raise SystemExit(f"all-reduce of ones over {world} ranks returned {probe.item()}")
    report["all_reduce_of_ones"] = int(probe.item())
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ragdolls", type=int, default=15000, help="ragdolls per GPU (15000 ~ 1.005M constraints)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lattice", action="store_true", help="BASELINE.json configs[4]: ONE connected ragdoll lattice of --ragdolls ragdolls split across the ranks "
                    "(strong scaling, boundary-velocity exchange after every pass) instead of the default independent islands per rank")
    ap.add_argument("--lattice-exact", action="store_true", help="with --lattice: the exact mode (bit-identical to one GPU: the configs[4] mode) — a device group on the island schedule, "
                    "one launch and one collective per step — instead of the per-pass block-Jacobi approximation (9 %% velocity error at the cut: outside BASELINE.json's tolerance)")
    ap.add_argument("--lattice-per-batch", action="store_true", help="with --lattice --lattice-exact: round 2's exact mode (an exchange after every batch on the launch-per-batch schedule) "
                    "instead of the device group on the island schedule")
    ap.add_argument("--lattice-no-clusters", action="store_true", help="with --lattice: the launch-per-batch schedule also in the per-pass mode (round 2's path)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the clock pre-warm of the setup phase (300 untimed solves, state restored afterwards)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 FETCH_SIZE/WRITE_SIZE child runs behind roofline.traffic")
    ap.add_argument("--traffic-child", default=None, choices=["main", "pile", "crowd"], help=argparse.SUPPRESS)
    ap.add_argument("--no-connected-scenes", action="store_true", help="skip the extra legs on connected scenes (100k-box pile = configs[1]; ragdoll crowd)")
    ap.add_argument("--full-report", default=None, help="where the long form of the bench line goes (default gpurun_out/bench_full.json); stdout carries the compact line")
    ap.add_argument("--no-scale-sweep", action="store_true", help="skip the scale_sweep leg (the headline scene at 1x / 2x / 4x / 8x the ragdolls)")
    ap.add_argument("--legs-in-process", action="store_true", help="run the extra legs in this process instead of one child process each (a leg that aborts then takes the JSON line with it)")
    ap.add_argument("--leg-child", default=None, choices=list(EXTRA_LEGS), help=argparse.SUPPRESS)
    ap.add_argument("--leg-out", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if not (args.traffic_child or args.leg_child):
        device_count = 0
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            import torch
            device_count = torch.cuda.device_count()
        action, detail = launch_plan(args.gpus, dict(os.environ), device_count, sys.argv[1:])
        if action == "error":
            raise SystemExit(f"bench.py: {detail}")
        if action == "spawn":  # (never an N = 1 line for an N > 1 request)
            raise SystemExit(subprocess.call(detail, env=dict(os.environ, MASTER_ADDR="127.0.0.1")))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not (args.traffic_child or args.leg_child):
        claim_stdout()  # (every rank: only rank 0 ever emits, and only the line)

    if args.traffic_child:
        return traffic_child(args, local_rank)
    if args.leg_child:
        import torch
        if not torch.cuda.is_available():  # (also brings torch's device state up before the library's, the order the parent process has: the counters' pass asks torch for the CU count)
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.init()
        from bepuphysics2_amd import build
        build.build_all()
        report = optional_leg(run_extra_leg, args.leg_child, args, int(os.environ.get("BEPU_BENCH_LEG_DEVICE", "0")))
        with open(args.leg_out, "w") as f:
            json.dump(report, f)
        return

    import torch
    dist = None
    self_checks = None
    if world > 1 or os.environ.get("BEPU_BENCH_FORCE_DIST") == "1":  # (the env switch lets a 1-GPU box exercise the RCCL code path)
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        self_checks = multi_gpu_self_checks(torch, dist, world, local_rank) if world > 1 else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")

    from bepuphysics2_amd import build
    if rank == 0:
        build.build_all()  # no-op when the in-tree .so files are current; one rank only: the outputs are shared files
    if dist is not None:
        dist.barrier()
    if args.lattice:
        return run_lattice(args, rank, local_rank, world, dist, torch)
    from bepuphysics2_amd.native import HipSolver
    from bepuphysics2_amd.roofline import INTEGRATE_BYTES_PER_BODY, FINAL_BYTES_PER_BODY, scene_stage_bytes
    from bepuphysics2_amd.scene import PoseIntegratorCallbacks, TYPE_TABLE

    from bepuphysics2_amd import sharding
    # configs[3]: ONE scene of world x ragdolls independent islands (weak scaling: the scene grows with the job), built identically on every rank and cut
    # into whole islands per rank by the partitioner the gloo test checks (tests/test_distributed_cpu.py: union of shares == single-process, bit for bit).
    scene, sd = build_scene(args.ragdolls * world, 5)
    whole_constraints = scene.constraint_count
    if world > 1 or os.environ.get("BEPU_BENCH_FORCE_DIST") == "1":
        scene = sharding.split_scene_by_islands(scene, world, only_rank=rank)[0].scene
    cb = PoseIntegratorCallbacks()
    dt = 1.0 / 60.0
    solver = HipSolver(device=local_rank, use_graph=not args.no_graph)
    solver.upload(scene)  # inputs resident in HBM before the timed region starts
    unit_state = specialise(solver)  # (setup: the island kernel for exactly this scene's twelve constraint types, prebuilt into csrc/units/)
    per_step_iterations = scene.constraint_count * int((1 + sd.iterations()).sum())
    # Part of the setup, disclosed in config.device_prewarm: the shader clock of an idle MI355X takes tens of milliseconds of load to reach its sustained
    # value (2.0 -> 2.2 GHz here). Run the same solve for a while, then restore the uploaded state device-to-device, so that the W warm-up steps and the K
    # timed steps below start from exactly the scene that was uploaded, on a device that is already clocked as it would be in a running simulation.
    prewarm_steps = 0 if args.no_prewarm else 300
    quiet_gc()  # (no full collection of Python's garbage collector inside the timed region: quiet_gc)
    for _ in range(prewarm_steps):
        solver.solve(dt, sd, cb, asynchronous=True)
    if prewarm_steps:
        solver.reset_state()
        solver.sync()

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
    kernel_family = solver.kernel_family()

    # ---- roofline of the dominant kernel: instrumented pass, HIP events on the solver's own stream around every launch ----
    roofline = None
    if rank == 0:
        ws_bytes, sv_bytes, inc_bytes = scene_stage_bytes(scene)
        its = sd.iterations()
        step_bytes = (sv_bytes * int(its.sum()) + ws_bytes * sd.substep_count + inc_bytes * (sd.substep_count - 1)
                      + INTEGRATE_BYTES_PER_BODY * scene.body_count * sd.substep_count + FINAL_BYTES_PER_BODY * scene.body_count)
        solver.set_profiling(True)
        prof_steps = max(3, min(40, args.steps))  # (forty event-bracketed launches: ten gave an average that moved by 3 % from run to run)
        agg = {}
        for _ in range(prof_steps):
            solver.set_profiling(False)
            for _ in range(20):  # keep the device busy right up to the instrumented step: an idle gap lets the shader clock drop and the event-bracketed
                solver.solve(dt, sd, cb, asynchronous=True)  # launch would then be timed at a clock the timed region above never sees
            solver.set_profiling(True)
            solver.solve(dt, sd, cb)
            for k, (ms, n) in solver.profile().items():
                a = agg.setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += n
        solver.set_profiling(False)
        families = {k: v[0] / prof_steps for k, v in agg.items()}
        step_gbs = step_bytes / (elapsed / args.steps) / 1e9
        if agg.get("cluster", [0.0, 0])[1] > 0:
            # Island-per-workgroup schedule: ONE launch runs every stage of every substep, so the units one launch processes are all
            # constraint-iterations of the step and its algorithmic bytes are the whole step's (SURVEY.md 8d figure x units).
            ms, n = agg["cluster"]
            avg_us = 1e3 * ms / n
            cyc = solver.cluster_cycles()
            detail = measure_traffic(args) if (not args.no_traffic and world == 1) else None  # the PMC child runs are an N=1 leg
            traffic = detail.get("bytes_per_launch") if isinstance(detail, dict) else None
            roofline = hbm_roofline("cluster_kernel (whole substep loop of a step in one launch)", avg_us, traffic, step_bytes, compulsory_stream_bytes(scene, sd),
                                    launches=n, traffic_detail=detail, working_set_bytes=working_set_bytes(scene),
                                    working_set_note="bodies + constraint rows; below the 256 MiB Infinity Cache FETCH_SIZE / WRITE_SIZE count fabric-side traffic that the "
                                                     "cache may serve — see scale_sweep for sizes that leave it",
                                    cluster_shader_kcycles_mean_max=[float(cyc.mean()) / 1e3, float(cyc.max()) / 1e3] if cyc.size else None,
                                    effective_shader_GHz=float(cyc.max()) / (avg_us * 1e3) if cyc.size else None,
                                    families_ms_per_step=families, step_algorithmic_GBs=step_gbs)
        else:
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
                        "families_ms_per_step": families, "other": fam["warmstart" if dom == "solve" else "solve"],
                        "step_algorithmic_GBs": step_gbs}
        if roofline is not None and "traffic_detail" not in roofline and not args.no_traffic and world == 1:  # launch-per-batch: counters of the dominant batch kernel
            detail = measure_traffic(args)
            roofline["traffic"] = detail.get("bytes_per_launch") if isinstance(detail, dict) else None
            roofline["traffic_detail"] = detail
            if roofline["traffic"]:  # same rule as the island schedule: frac is the counters' bandwidth fraction, the algorithmic figure keeps its own keys
                roofline["algorithmic_GBs"], roofline["algorithmic_frac_of_peak"] = roofline["achieved"], roofline["frac"]
                roofline["achieved"] = roofline["traffic"] / (roofline["avg_launch_us"] * 1e-6) / 1e9
                roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
                roofline["basis"] = "pmc"

    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        baseline = optional_leg(cpu_baseline, args.ragdolls, 5)  # the same scene as the GPU leg; bounded by time (about 12 s after picking the thread count)

    main_clustered = bool(solver.cluster_cycles().size)
    row_policy = {-1: "still measuring", 0: "plain constraint-row accesses", 1: "non-temporal constraint-row accesses", 2: "plain rows + one 8 KB span of code touched ahead per work item"}[solver.row_policy()] if main_clustered else None
    connected = None
    def extra(name):  # (every extra leg in a process of its own unless --legs-in-process: leg_in_child)
        return optional_leg(run_extra_leg, name, args, local_rank) if args.legs_in_process else leg_in_child(name, args, local_rank)

    if rank == 0 and world == 1 and not args.no_connected_scenes and not args.traffic_child:
        solver.close()
        connected = {"pile_100k": extra("pile"), "ragdoll_crowd": extra("crowd")}

    boundary = None
    lattice_report = None
    widened = None
    sweep = None
    if rank == 0 and world == 1 and not args.no_scale_sweep and not args.no_connected_scenes and not args.traffic_child:
        solver.close()
        sweep = extra("sweep")
    if rank == 0 and world == 1 and not args.no_connected_scenes and not args.traffic_child:
        widened = extra("widened")
        boundary = extra("boundary")
        lattice_report = extra("lattice")

    lattice_group = None
    if dist is not None and (world > 1 or os.environ.get("BEPU_BENCH_GROUP_LEG") == "1") and not args.no_connected_scenes:
        # BASELINE.json configs[4] beside configs[3] on ONE line at N > 1 (VERDICT r5 next #7b): the same number of ragdolls as one rank's share, as ONE connected lattice
        # split over the N devices as a device group (exact mode). Strong scaling inside this leg: its own ms_per_step, and the N = 1 figure of the same lattice is the
        # `lattice.device_group_single_rank_ms` of an N = 1 run. The device group has never run on more than one GPU (the pool has single-GPU boxes), so the leg runs the
        # way the N = 1 legs do — every rank starts a child of its own, the children meet on a rendezvous of their own — and a child that aborts or hangs costs the
        # leg's report, not the headline's line (lattice_group_in_children).
        solver.close()
        lattice_group = lattice_group_in_children(args, rank, local_rank, world, dist)
    if rank == 0:
        value = whole_job_rate
        out = {
            "metric": "constraint-iterations/sec", "value": value, "unit": "constraint-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RagdollTubeBenchmark scaled to ~1M constraints (BASELINE.json configs[2]): "
                                   f"{args.ragdolls} ragdolls/GPU, {scene.constraint_count} constraints/GPU, {scene.body_count} bodies/GPU, "
                                   f"{len(scene.batches)} batches, {sd.substep_count} substeps x {sd.velocity_iteration_count} velocity iteration(s), dt=1/60",
                       "sharding": (f"ONE scene of {args.ragdolls * world} ragdolls ({whole_constraints} constraints) cut into whole islands per GPU by "
                                    "sharding.split_scene_by_islands (BASELINE.json configs[3]); no data-path collective, RCCL barrier + timing reduction only")
                                   if (world > 1 or dist is not None) else "single GPU",
                       "schedule": "island-per-workgroup: one plain kernel launch per step" if main_clustered else
                                   ("launch-per-batch" + ("" if args.no_graph else ", hipGraph replay")),
                       "row_policy": row_policy and f"{row_policy}, picked from the timings of the first fifteen solves after the upload (all candidates bit-identical; DESIGN.md 5)",
                       "device_prewarm": f"{prewarm_steps} untimed solves during setup, uploaded state restored before the {args.warmup} warm-up steps",
                       "kernel_family": {-1: "none (launch-per-batch)", 0: "contacts family (8 types)", 1: "hot family (16 types)", 2: "wide family (44 types)",
                                         3: "a unit compiled for exactly the scene's constraint types (bepuhip_specialise_units; prebuilt by build() into csrc/units/)"}.get(kernel_family, str(kernel_family)),
                       "finite": finite},
            "roofline": roofline, "cpu_baseline": baseline, "connected_scenes": connected, "scale_sweep": sweep, "widened_types": widened, "boundary": boundary, "lattice": lattice_report,
            "lattice_device_group": lattice_group, "self_checks": self_checks,
        }
        if isinstance(baseline, dict) and baseline.get("value"):
            baseline["gpu_over_cpu"] = value / baseline["value"]
            if (baseline.get("ideal_socket_bound") or {}).get("value"):
                baseline["gpu_over_ideal_socket"] = value / baseline["ideal_socket_bound"]["value"]  # the figure to hold against north_star's >= 10x
        full_path = args.full_report or os.path.join(REPO, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f)
        except OSError:
            full_path = None
        emit_line(compact_line(out, full_path))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    solver.close()


if __name__ == "__main__":
    main()
