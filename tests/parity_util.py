"""Shared helpers: run the same seeded scene through the oracle (CPU) and the HIP path (through the C ABI) and compare."""
from __future__ import annotations

import numpy as np

import oracle_ffi


def run_oracle(scene, dt, sd, cb, frames=1, threads=1):
    s = scene.copy()
    for _ in range(frames):
        oracle_ffi.solve(s, dt, sd, cb, threads=threads)
    return s


def run_hip(solver, scene, dt, sd, cb, frames=1):
    s = scene.copy()
    solver.upload(s, sd.fallback_batch_threshold)
    for _ in range(frames):
        solver.solve(dt, sd, cb)
    solver.download(s)
    return s


def max_ulp_diff(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.abs(a - b).max()) if a.size else 0


def compare_scenes(ref, got, rel_tol=1e-4):
    """Returns dict of metrics. Bodies: pose floats 0-6, velocity 8-10,12-14. Padding floats are ignored."""
    cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]
    rb, gb = ref.bodies[:, cols], got.bodies[:, cols]
    out = {"bodies_bit_exact": bool(np.array_equal(rb.view(np.int32), gb.view(np.int32))), "bodies_max_ulp": max_ulp_diff(rb, gb)}
    vel_ref = ref.bodies[:, [8, 9, 10, 12, 13, 14]]
    vel_got = got.bodies[:, [8, 9, 10, 12, 13, 14]]
    denom = max(float(np.abs(vel_ref).max()), 1e-6)
    out["velocity_rel_err"] = float(np.abs(vel_ref - vel_got).max() / denom)
    pos_ref, pos_got = ref.bodies[:, 4:7], got.bodies[:, 4:7]
    out["position_rel_err"] = float(np.abs(pos_ref - pos_got).max() / max(float(np.abs(pos_ref).max()), 1e-6))
    imp_exact, pre_exact, imp_ulp = True, True, 0
    for br, bg in zip(ref.batches, got.batches):
        for tr, tg in zip(br, bg):
            assert tr.type_id == tg.type_id and tr.count == tg.count
            occupied = tr.occupied(ref.bundle_width)  # the empty lanes of a sequential-fallback type batch hold nothing that is ever read
            ar, ag = tr.accumulated_lanes(ref.bundle_width)[occupied], tg.accumulated_lanes(got.bundle_width)[occupied]
            pr, pg = tr.prestep_lanes(ref.bundle_width)[occupied], tg.prestep_lanes(got.bundle_width)[occupied]
            imp_exact &= bool(np.array_equal(ar.view(np.int32), ag.view(np.int32)))
            pre_exact &= bool(np.array_equal(pr.view(np.int32), pg.view(np.int32)))
            imp_ulp = max(imp_ulp, max_ulp_diff(ar, ag))
    out["impulses_bit_exact"] = imp_exact
    out["prestep_bit_exact"] = pre_exact
    out["impulses_max_ulp"] = imp_ulp
    return out


def compare_scenes_with_nans(ref, got):
    """compare_scenes for scenes that are MEANT to hold NaN / infinity: a word is NaN in both or in neither, and every word that is not NaN is bit-equal.
    Which NaN it is (sign, payload) is the hardware's business — x86 makes the negative 'real indefinite' out of inf - inf, gfx950 the positive one — where it appears is not:
    that is decided by the order of the operands of Vector.Min / Max (minps / maxps return the SECOND operand when either is NaN) and by the comparisons' outcomes on NaN.
    Bodies (padding floats ignored), accumulated impulses and prestep rows; returns the NaN word counts so that a caller can insist the scene really held some."""
    cols = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13, 14]

    def same(a, b):
        na, nb = np.isnan(a), np.isnan(b)
        return bool(np.array_equal(na, nb)) and bool(np.array_equal(a.view(np.int32)[~na], b.view(np.int32)[~nb])), int(na.sum())

    out = {}
    out["bodies_same"], out["body_nans"] = same(np.ascontiguousarray(ref.bodies[:, cols]), np.ascontiguousarray(got.bodies[:, cols]))
    out["body_infinities"] = int(np.isinf(ref.bodies[:, cols]).sum())
    imp, pre, nans = True, True, 0
    for br, bg in zip(ref.batches, got.batches):
        for tr, tg in zip(br, bg):
            assert tr.type_id == tg.type_id and tr.count == tg.count
            occupied = tr.occupied(ref.bundle_width)
            ok, n = same(np.ascontiguousarray(tr.accumulated_lanes(ref.bundle_width)[occupied]), np.ascontiguousarray(tg.accumulated_lanes(got.bundle_width)[occupied]))
            imp &= ok; nans += n
            ok, n = same(np.ascontiguousarray(tr.prestep_lanes(ref.bundle_width)[occupied]), np.ascontiguousarray(tg.prestep_lanes(got.bundle_width)[occupied]))
            pre &= ok; nans += n
    out["impulses_same"], out["prestep_same"], out["constraint_nans"] = imp, pre, nans
    return out
